"""Fixed-effect fit on one MI355X: time per objective evaluation and achieved HBM bandwidth.

    PYTHONPATH=. python tools/fe_bench.py [rows] [nnz_per_row] [features]
Algorithmic bytes per evaluation: CSR pass 8 B/nnz (fp32 value + int32 column) + CSC pass 8 B/nnz (value + row) +
16 B/row (row pointer, y, offset, weight) + 8 B/row residual written and gathered + per coefficient: x read, g written,
d, r and 2m history vectors read for the fused dot products = (4 + 2m) * 8 B.
"""
import json
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
cols = rng.integers(0, D, (n, k), dtype=np.int64)
vals = rng.standard_normal((n, k)).astype(np.float32)
w_true = (rng.standard_normal(D) * 0.1)
z = (vals * w_true[cols]).sum(1)
y = (rng.random(n) < 1 / (1 + np.exp(-z))).astype(np.float32)
off = np.zeros(n, np.float32)
rp = np.arange(n + 1, dtype=np.int64) * k
s = REDeviceSolver(0)
s.set_timing(True)
batch, _ = fe.shard_as_batch(rp, cols.ravel(), vals.ravel(), y, off, None, True)
t0 = time.perf_counter()
raw = s.upload(batch)
torch.cuda.synchronize()
t_up = time.perf_counter() - t0
t0 = time.perf_counter()
packed = s.pack(raw)
torch.cuda.synchronize()
t_pack = time.perf_counter() - t0
out = s.alloc_result(packed)
for max_iter in (30, 30):
    opts = SolverOptions(l2=1.0, regularize_bias=True, has_intercept=True, m=10, max_iter=max_iter, threshold=0.0, sum_loss=True)
    t0 = time.perf_counter()
    res = s.solve(packed, opts, out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
h = res.to_host()
nfev, nit = int(h["nfev"][0]), int(h["nit"][0])
Z = n * k
p = packed.P
m = 10
bytes_eval = 16.0 * Z + 16.0 * n + 16.0 * n + (4 + 2 * m) * 8.0 * p
ms_eval = dt * 1e3 / nfev
line = {"metric": "fixed-effect objective+gradient evaluations/sec (one L-BFGS evaluation incl. direction update)",
        "rows": n, "nnz": Z, "features": D, "coefficients": int(p), "nit": nit, "nfev": nfev, "status": int(h["status"][0]),
        "fit_ms": dt * 1e3, "ms_per_evaluation": ms_eval, "evaluations_per_s": 1e3 / ms_eval,
        "alg_bytes_per_evaluation": bytes_eval, "achieved_GBps": bytes_eval / (ms_eval * 1e-3) / 1e9,
        "hbm_peak_GBps": 8000.0, "frac": bytes_eval / (ms_eval * 1e-3) / 8e12, "pack_ms": t_pack * 1e3, "upload_ms": t_up * 1e3}
print(json.dumps(line))
