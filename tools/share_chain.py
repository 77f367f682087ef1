"""Which entities are the longest chains of a share's one-wavefront tall class? (samples x evaluations per entity)

    PYTHONPATH=. python tools/share_chain.py ml20m_user 3
"""
import sys

import numpy as np
import torch

import bench_strong
from gdmix_amd.solver import REDeviceSolver, SolverOptions

name = sys.argv[1] if len(sys.argv) > 1 else "ml20m_user"
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s = REDeviceSolver(0)
s.set_timing(True)
share = bench_strong.make_share(name, 8, rank, s, 0)
opts = SolverOptions(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
packed = s.pack(share.raw_dev)
res = s.solve(packed, opts)
torch.cuda.synchronize()
ms = np.array(s.last_solve_ms())
cls = packed._view(packed.c.cls_tmp, packed.E, torch.int32).cpu().numpy()
names = [n for n, _ in s.class_counts(packed)]
nfev = res.nfev.cpu().numpy()
nit = res.nit.cpu().numpy()
n = share.n
for c in np.flatnonzero(ms > 0):
    sel = np.flatnonzero(cls == c)
    if sel.size == 0:
        continue
    work = n[sel] * nfev[sel]
    top = sel[np.argsort(-nfev[sel])[:5]]
    topw = sel[np.argsort(-work)[:5]]
    print(f"{names[c]:44s} {sel.size:6d} entities {ms[c]:7.3f} ms | nfev mean {nfev[sel].mean():5.1f} max {nfev[sel].max():4d} | most evaluations (n, nfev): "
          f"{[(int(n[e]), int(nfev[e])) for e in top]} | most samples x evaluations: {[(int(n[e]), int(nfev[e])) for e in topw]}")
