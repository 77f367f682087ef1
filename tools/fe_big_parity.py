"""One large fixed-effect shard against the CPU oracle (beyond the sizes tests/ and tools/fuzz_fe.py draw): rows in the millions,
Zipf-distributed columns, units that span more than 2^21 gathered elements (the three-array form of the entries).
    PYTHONPATH=.:tests python tools/fe_big_parity.py [rows] [nnz_per_row] [features]"""
import sys
import time

import numpy as np

from gdmix_amd import fixed_effect as fe
from gdmix_amd.solver import REDeviceSolver
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
D = int(sys.argv[3]) if len(sys.argv) > 3 else 300_000
rng = np.random.default_rng(3)
cols = np.minimum((float(D + 1) ** rng.random((n, k))).astype(np.int64) - 1, D - 1).ravel()
vals = (rng.standard_normal(n * k) * 0.5).astype(np.float32)
w_true = rng.standard_normal(D) * 0.3
z = (vals.astype(np.float64) * w_true[cols]).reshape(n, k).sum(1)
y = (rng.random(n) < 1 / (1 + np.exp(-z))).astype(np.float32)
off = (0.1 * rng.standard_normal(n)).astype(np.float32)
wt = (0.5 + rng.random(n)).astype(np.float32)
rp = np.arange(n + 1, dtype=np.int64) * k
s = fe.FixedEffectDeviceSolver(solver=REDeviceSolver(0))
kw = dict(offset=off, weight=wt, l2=5.0, regularize_bias=False, max_iter=12)
t = time.perf_counter()
th, info = s.fit_stepping(rp, cols, vals, y, D, **kw)
print(f"device: {time.perf_counter() - t:.2f} s  status {info['status']} nit {info['nit']} nfev {info['nfev']}")
batch, dummy = fe.shard_as_batch(rp, cols, vals, y, off, wt, True)
pk = oracle.pack(batch.ent_row_ptr, batch.row_nnz_ptr, batch.col_global)
o = oracle.make_opts(l2=5.0, regularize_bias=False, has_intercept=True, max_iter=12, threshold=0.0, sum_loss=True)
t = time.perf_counter()
res = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o)
th_o = fe.to_global(res["theta"], pk["unique_global"], D, True, dummy)
err = float(np.max(np.abs(th - th_o)) / np.max(np.abs(th_o)))
print(f"oracle: {time.perf_counter() - t:.2f} s  status {res['status'][0]} nit {res['nit'][0]} nfev {res['nfev'][0]}   theta rel err {err:.3e}")
assert info["status"] == res["status"][0] and info["nit"] == res["nit"][0] and err < 1e-7
print("ok")
