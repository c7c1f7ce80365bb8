import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from cuclarabel_amd import problems
from cuclarabel_amd.kktsolver import HipKKTSolver
n=int(sys.argv[1]) if len(sys.argv)>1 else 20000
pb=problems.config2(n=n)
t=time.time(); ks=HipKKTSolver(pb.P,pb.A,pb.cones); print('setup',time.time()-t, ks.info, flush=True)
ks.profile_enable(True)
rng=np.random.default_rng(0); rx,rz=rng.standard_normal(pb.n),rng.standard_normal(pb.m)
x,z=np.zeros(pb.n),np.zeros(pb.m)
for it in range(5):
    t=time.time(); ok=ks.kktsolver_update_from_sz(pb.s0,pb.z0); t1=time.time()-t
    ks.kktsolver_setrhs(rx,rz); t=time.time(); ok2=ks.kktsolver_solve(x,z); t2=time.time()-t
    print(it, ok, ok2, 'update %.2f ms solve %.2f ms ir=%d'%(t1*1e3,t2*1e3,ks.last_ir_iterations), flush=True)
print(ks.profile())
