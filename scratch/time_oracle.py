import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/scratch')
import numpy as np, ctypes as C
from cuclarabel_amd import problems
from tests.oracle_bindings import make_oracle
n=int(sys.argv[1])
pb=problems.config2(n=n)
# AMD perm from product symbolic (scratch lib)
lib=C.CDLL('/root/repo/scratch/libperm.so')
o0=make_oracle(pb, perm=np.arange(pb.n+pb.m+2*(n//100)))
K=o0.K(); N=K.shape[0]
perm=np.zeros(N,dtype=np.int64)
i64p=np.ctypeslib.ndpointer(dtype=np.int64)
lib.get_perm.argtypes=[C.c_int,i64p,i64p,C.c_int,i64p]
for ordn in (0,1):
    lib.get_perm(N,K.indptr.astype(np.int64),K.indices.astype(np.int64),ordn,perm)
    o=make_oracle(pb, perm=perm.copy())
    o.update_scaling(pb.s0,pb.z0)
    t=time.time(); o.kktsolver_update(); tf=time.time()-t
    rng=np.random.default_rng(0); o.kktsolver_setrhs(rng.standard_normal(pb.n),rng.standard_normal(pb.m))
    t=time.time(); o.kktsolver_solve(); ts=time.time()-t
    b=rng.standard_normal(N); t=time.time(); o.ldl_solve(b); tl=time.time()-t
    print('order',ordn,'nnzL',o.nnzL,'update+factor %.1f ms  solve(IR=%d) %.1f ms  bare trisolve %.1f ms'%(tf*1e3,o.last_ir_iters,ts*1e3,tl*1e3))
