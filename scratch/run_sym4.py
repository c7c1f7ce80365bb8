import sys; sys.path.insert(0,'/root/repo/scratch'); sys.path.insert(0,'/root/repo')
from run_sym import *
pb=problems.config2(n=100000)
run(pb,1,1000)
