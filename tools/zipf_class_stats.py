"""Per size-class statistics of the Zipf workload: entity shape, evaluations, and the bytes one evaluation moves
(matrix copies and L-BFGS history), so the tail kernels can be priced against HBM."""
import sys

import numpy as np
import torch

from gdmix_amd import synthetic
from gdmix_amd.solver import REDeviceSolver, SolverOptions

E = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
b = synthetic.make_batch(E, 32, 8, 65536, seed=synthetic.C5_SEED, size_dist="zipf", with_uid=False)
s = REDeviceSolver(0)
s.set_timing(True)
o = SolverOptions(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
raw = s.upload(b)
pk = s.pack(raw)
out = s.alloc_result(pk)
for _ in range(2):
    res = s.solve(pk, o, out=out)
ms = np.array(s.last_solve_ms())
classes = s.class_counts(pk)
cls = pk._view(pk.c.cls_tmp, pk.E, torch.int32).cpu().numpy()
n = b.ent_n().astype(np.float64)
z = b.ent_nnz().astype(np.float64)
p = np.diff(pk.coef_ptr_host()).astype(np.float64)
nfev = res.nfev.cpu().numpy().astype(np.float64)
nit = res.nit.cpu().numpy().astype(np.float64)
st = res.status.cpu().numpy()
print("%-44s %8s %9s %8s %8s %6s %6s %9s %10s %10s %8s" % ("class", "count", "nnz", "n", "p", "nfev", "nit", "ms", "mat GB/s", "hist GB/s", "maxit%"))
for c, (name, cnt) in enumerate(classes):
    if not cnt:
        continue
    k = cls == c
    mat = (nfev[k] * (16.0 * z[k] + 24.0 * n[k])).sum()                  # CSR + CSC copies per evaluation
    hist = (nit[k] * (2 * 10 * p[k] * 8.0 * 2 + 6 * p[k] * 8.0)).sum()    # S,Y read for products + direction, vectors
    print("%-44s %8d %9.0f %8.0f %8.0f %6.1f %6.1f %9.3f %10.1f %10.1f %8.2f" % (
        name[:44], cnt, z[k].mean(), n[k].mean(), p[k].mean(), nfev[k].mean(), nit[k].mean(), ms[c],
        mat / ms[c] / 1e6, hist / ms[c] / 1e6, 100.0 * (st[k] == 2).mean()))
