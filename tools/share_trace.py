"""One share of a strongly scaled MovieLens-20M job, pack + solve a few times — to be run under rocprofv3 --kernel-trace: the
timeline of one step (which class launches overlap) is printed by tools/share_timeline.py from the trace.

    PYTHONPATH=. python tools/share_trace.py ml20m_movie 3 [steps [ranks]]        (ranks = 1: the whole population)
"""
import sys

import torch

import bench_strong
from gdmix_amd.solver import REDeviceSolver, SolverOptions

name = sys.argv[1] if len(sys.argv) > 1 else "ml20m_movie"
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ranks = int(sys.argv[4]) if len(sys.argv) > 4 else 8
s = REDeviceSolver(0)
share = bench_strong.make_share(name, ranks, rank, s, 0)
opts = SolverOptions(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
for i in range(steps):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    packed = s.pack(share.raw_dev)
    e1.record()
    res = s.solve(packed, opts)
    e2.record()
    torch.cuda.synchronize()
    print(f"step {i}: pack {e0.elapsed_time(e1):.3f} ms, solve {e1.elapsed_time(e2):.3f} ms")
