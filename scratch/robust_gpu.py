import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from cuclarabel_amd import problems
from cuclarabel_amd.kktsolver import HipKKTSolver
from tests.oracle_bindings import make_oracle
def run(name, pb, use_sz=True):
    t=time.time(); ks=HipKKTSolver(pb.P,pb.A,pb.cones); ts=time.time()-t
    o=make_oracle(pb, perm=ks.perm())
    assert o.update_scaling(pb.s0,pb.z0) and o.kktsolver_update()
    if use_sz: ok=ks.kktsolver_update_from_sz(pb.s0,pb.z0)
    else:
        u,v,e2,_=o.soc_sparse(); ok=ks.kktsolver_update(o.get_Hs(),u,v,e2)
    rng=np.random.default_rng(0); rx,rz=rng.standard_normal(pb.n),rng.standard_normal(pb.m)
    ks.kktsolver_setrhs(rx,rz); o.kktsolver_setrhs(rx,rz)
    x,z=np.zeros(pb.n),np.zeros(pb.m); ok2=ks.kktsolver_solve(x,z)
    _,xo,zo=o.kktsolver_solve()
    sc=max(abs(xo).max(),abs(zo).max()); err=max(abs(x-xo).max(),abs(z-zo).max())/sc
    i=ks.info
    print(f"{name}: N={i['N']} nnzL={i['nnzL']} maxfront={i['max_front']} levels={i['nlevels']} setup={ts:.2f}s ok={ok},{ok2} ir={ks.last_ir_iterations}/{o.last_ir_iters} relerr={err:.2e}", flush=True)
run('cfg3 10x300', problems.config3(nblocks=10, blk=300))
run('cfg3 4x500', problems.config3(nblocks=4, blk=500))
run('cfg5 small', problems.config5(n=600, npsd=12, psd_dim=10, nsoc=8, soc_dim=30), use_sz=False)
run('cfg5 psd20', problems.config5(n=1000, npsd=20, psd_dim=20, nsoc=10, soc_dim=50), use_sz=False)
run('cfg4 j0', problems.config4(j=0))
run('cfg2u', problems.config_unstructured(n=3000))
