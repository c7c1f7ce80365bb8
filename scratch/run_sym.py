import sys, ctypes as C, numpy as np, time
sys.path.insert(0,'/root/repo')
from cuclarabel_amd import problems
from tests.oracle_bindings import make_oracle
lib=C.CDLL('/root/repo/scratch/libsym.so')
i64p=np.ctypeslib.ndpointer(dtype=np.int64); f64p=np.ctypeslib.ndpointer(dtype=np.float64)
lib.sym_stats.argtypes=[C.c_int,i64p,i64p,C.c_int,C.c_int,f64p,C.c_int,C.c_int,C.c_int,C.c_double,C.c_double,C.c_double]
def run(pb, ordering, leaf=200, relax=(8,32,128,0.5,0.15,0.05)):
    o=make_oracle(pb, perm=np.arange(pb.n+pb.m+2*sum(1 for c in pb.cones if c.kind==2 and c.dim>4)))
    K=o.K()
    out=np.zeros(8)
    lib.sym_stats(K.shape[0],K.indptr.astype(np.int64),K.indices.astype(np.int64),ordering,leaf,out,*relax)
    return out
if __name__=='__main__':
    cfg=sys.argv[1]; n=int(sys.argv[2]) if len(sys.argv)>2 else None
    pb={'1':lambda:problems.config1(),'2':lambda:problems.config2(n=n or 100000),'3':lambda:problems.config3(),'5':lambda:problems.config5(),'2u':lambda:problems.config_unstructured()}[cfg]()
    for ordn,name in ((0,'AMD'),(1,'ND')):
        print('==',name); run(pb,ordn)
