"""Fixed-effect fit on one MI355X: time per objective evaluation and achieved HBM bandwidth.

    PYTHONPATH=. python tools/fe_bench.py [rows] [nnz_per_row] [features] [uniform|zipf]
(zipf: feature j drawn with probability ~ 1/(j+1), the shape of real sparse features: the most frequent one holds 1/ln(features)
of all entries)
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
dist = sys.argv[4] if len(sys.argv) > 4 else "uniform"
rng = np.random.default_rng(0)
if dist == "zipf":
    cols = np.minimum((float(D + 1) ** rng.random((n, k))).astype(np.int64) - 1, D - 1)
else:
    cols = rng.integers(0, D, (n, k), dtype=np.int64)
vals = rng.standard_normal((n, k)).astype(np.float32)
w_true = (rng.standard_normal(D) * 0.1)
z = (vals * w_true[cols]).sum(1)
y = (rng.random(n) < 1 / (1 + np.exp(-z))).astype(np.float32)
off = np.zeros(n, np.float32)
rp = np.arange(n + 1, dtype=np.int64) * k
s = REDeviceSolver(0)
fes = fe.FixedEffectDeviceSolver(solver=s)
results = {}
import os
for label in os.environ.get("FE_BENCH_PATHS", "stepping,team").split(","):
    fit = fes.fit_stepping if label == "stepping" else fes.fit
    for max_iter in (30, 30):   # second run is the measured one
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fit(rp, cols.ravel(), vals.ravel(), y, D, offset=off, l2=1.0, regularize_bias=True, max_iter=max_iter,
                  **({"return_problem": True} if label == "stepping" else {}))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    theta, info = out[0], out[1]
    extra = {}
    if label == "stepping":
        prob = out[2]
        # time the device part alone: evaluations + steps of a fresh problem without the host-side pack
        prob.close()
        batch, _ = fe.shard_as_batch(rp, cols.ravel(), vals.ravel(), y, off, None, True)
        packed = s.pack(batch)
        from gdmix_amd.solver import SolverOptions
        opts = SolverOptions(l2=1.0, regularize_bias=True, has_intercept=True, m=10, max_iter=30, threshold=0.0, sum_loss=True)
        p2 = fe._SteppingProblem(s, packed, D, opts, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = fe.run_stepping_loop(p2)
        torch.cuda.synchronize()
        dt_dev = time.perf_counter() - t0
        th2, info2 = p2.result()
        rows_ms, cols_ms = p2.last_eval_ms()
        extra = {"device_loop_ms": dt_dev * 1e3, "rows_pass_ms": rows_ms, "cols_pass_ms": cols_ms, "nfev": info2["nfev"]}
        p2.close()
    results[label] = (dt, info, extra)
Z = n * k
m = 10
P = D + 1
for label, (dt, info, extra) in results.items():
    nfev = int(info["nfev"])
    bytes_eval = 16.0 * Z + 16.0 * n + 16.0 * n + (4 + 2 * m) * 8.0 * P
    dev_ms = extra.get("device_loop_ms", dt * 1e3)
    ms_eval = dev_ms / nfev
    line = {"path": label, "metric": "fixed-effect L-BFGS evaluations/sec (objective + gradient + step)",
            "rows": n, "nnz": Z, "features": D, "columns": dist, "nit": int(info["nit"]), "nfev": nfev, "status": int(info["status"]),
            "fit_wall_ms_incl_upload_pack": dt * 1e3, "ms_per_evaluation": ms_eval, "alg_bytes_per_evaluation": bytes_eval,
            "achieved_GBps": bytes_eval / (ms_eval * 1e-3) / 1e9, "hbm_peak_GBps": 8000.0,
            "frac": bytes_eval / (ms_eval * 1e-3) / 8e12, **extra}
    if "rows_pass_ms" in extra:
        line["rows_pass_GBps"] = (8.0 * Z + 24.0 * n) / (extra["rows_pass_ms"] * 1e-3) / 1e9
        line["cols_pass_GBps"] = (8.0 * Z + 8.0 * n) / (extra["cols_pass_ms"] * 1e-3) / 1e9
    print(json.dumps(line))
