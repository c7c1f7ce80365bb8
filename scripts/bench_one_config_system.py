#!/usr/bin/env python3
"""One configuration's unit through the reduced-system layer (level C, lazy: kkt_update!, kkt_solve!(:affine),
kkt_solve!(:combined)) a few times, for profiling: bench_one_config_system.py <cfg> [reps]."""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch
from cuclarabel_amd import problems
from cuclarabel_amd.kktsolver import HipKKTSolver, HipKKTSystem

pb = getattr(problems, "config" + sys.argv[1])()
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
ks = HipKKTSolver(pb.P, pb.A, pb.cones)
ks.set_stream(torch.cuda.current_stream(dev).cuda_stream)
system = HipKKTSystem(ks)
system.init(pb.q, pb.b)
system.set_lazy(True)
rng = np.random.default_rng(0)
dd = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
var = [dd(rng.standard_normal(pb.n)), dd(pb.s0), dd(pb.z0)]
rhs = [[dd(rng.standard_normal(k)) for k in (pb.n, pb.m, pb.m)] for _ in range(2)]
lhs = [torch.zeros(k, dtype=torch.float64, device=dev) for k in (pb.n, pb.m, pb.m)]
P = lambda ts: [t.data_ptr() for t in ts]
update, solve_affine, solve_combined = system.prepared(P(lhs), [P(rhs[0]), P(rhs[1])], 0.3, -0.1, P(var), 1.1, 0.9)
for _ in range(reps):
    assert update()
    assert solve_affine()[0] and solve_combined()[0]
torch.cuda.synchronize()
print("done", ks.info["nlevels"], ks.fallbacks)
