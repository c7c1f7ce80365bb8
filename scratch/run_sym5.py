import sys; sys.path.insert(0,'/root/repo/scratch'); sys.path.insert(0,'/root/repo')
from run_sym import *
pb=problems.config2(n=100000)
for relax in ((8,32,128,0.5,0.15,0.05),(8,32,128,0.6,0.3,0.15),(16,64,256,0.6,0.3,0.2),(16,64,256,0.8,0.5,0.3)):
    print('== relax',relax, flush=True); run(pb,1,1000,relax)
