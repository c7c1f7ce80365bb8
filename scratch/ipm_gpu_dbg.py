import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from cuclarabel_amd import problems, ipm
from tests.ipm_backends import OracleBackend
pb = problems.config2(n=1000)
res = ipm.solve(pb.P, pb.q, pb.A, pb.b, pb.cones, ipm.HipBackend(pb.P, pb.A, pb.cones))
ref = ipm.solve(pb.P, pb.q, pb.A, pb.b, pb.cones, OracleBackend(pb.P, pb.A, pb.cones))
print(res.status, res.iterations, res.kkt_ir_rounds, '|', ref.status, ref.iterations, ref.kkt_ir_rounds)
for a,b in zip(res.history, ref.history):
    print(a['iter'], '%.6e %.6e | %.3e %.3e | %.3e %.3e | mu %.3e %.3e'%(a['pcost'],b['pcost'],a['pres'],b['pres'],a['dres'],b['dres'],a['mu'],b['mu']))
