"""Where a fixed-effect fit spends its wall time on the device side: upload, pack (CSR + CSC of one giant entity), gdmix_fe_create
(the two passes' copies of the non-zeros), the L-BFGS loop.   PYTHONPATH=. python tools/fe_phases.py [rows] [nnz_per_row] [features]"""
import sys
import time

import numpy as np
import torch

from gdmix_amd import fixed_effect as fe
from gdmix_amd.solver import REDeviceSolver, SolverOptions

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
D = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
rng = np.random.default_rng(0)
cols = rng.integers(0, D, (n, k), dtype=np.int64).ravel()
vals = rng.standard_normal(n * k).astype(np.float32)
y = (rng.random(n) < 0.5).astype(np.float32)
off = np.zeros(n, np.float32)
rp = np.arange(n + 1, dtype=np.int64) * k
s = REDeviceSolver(0)
opts = SolverOptions(l2=1.0, regularize_bias=True, has_intercept=True, m=10, max_iter=30, threshold=0.0, sum_loss=True)


def timed(label, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print(f"  {label:34s} {1e3 * (time.perf_counter() - t):8.1f} ms", flush=True)
    return r


for rep in range(2):
    print("pass", rep)
    batch, _ = timed("shard_as_batch (host)", lambda: fe.shard_as_batch(rp, cols, vals, y, off, None, True))
    raw = timed("upload (pageable H2D)", lambda: s.upload(batch))
    packed = timed("gdmix_re_pack", lambda: s.pack(raw))
    prob = timed("gdmix_fe_create", lambda: fe._SteppingProblem(s, packed, D, opts, None))
    timed("L-BFGS loop (30 iterations)", lambda: fe.run_stepping_loop(prob))
    prob.close()
