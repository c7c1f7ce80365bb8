#!/usr/bin/env python3
"""One configuration's unit (update + refactor + 3 solves) a few times, for profiling: bench_one_config.py <cfg> [reps]."""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch
from cuclarabel_amd import problems
from cuclarabel_amd.kktsolver import HipKKTSolver

pb = getattr(problems, "config" + sys.argv[1])()
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
ks = HipKKTSolver(pb.P, pb.A, pb.cones)
ks.set_stream(torch.cuda.current_stream(dev).cuda_stream)
rng = np.random.default_rng(0)
s, z = torch.from_numpy(pb.s0).to(dev), torch.from_numpy(pb.z0).to(dev)
rx, rz = torch.from_numpy(rng.standard_normal(pb.n)).to(dev), torch.from_numpy(rng.standard_normal(pb.m)).to(dev)
lx, lz = torch.zeros(pb.n, dtype=torch.float64, device=dev), torch.zeros(pb.m, dtype=torch.float64, device=dev)
for _ in range(reps):
    assert ks.kktsolver_update_from_sz_dev(s.data_ptr(), z.data_ptr())
    for _ in range(3):
        ks.kktsolver_setrhs_dev(rx.data_ptr(), rz.data_ptr())
        assert ks.kktsolver_solve_dev(lx.data_ptr(), lz.data_ptr())
torch.cuda.synchronize()
print("done", ks.info["nlevels"])
