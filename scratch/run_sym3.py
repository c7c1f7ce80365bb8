import sys; sys.path.insert(0,'/root/repo/scratch'); sys.path.insert(0,'/root/repo')
from run_sym import *
pb=problems.config2(n=10000)
for leaf in (200,1000,4000,10000,40000):
    print('== ND leaf',leaf, flush=True); run(pb,1,leaf)
