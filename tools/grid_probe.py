"""Time the device-wide kernel per L-BFGS iteration on small entities (pure barrier latency) and on one giant."""
import sys
import numpy as np
import torch
from gdmix_amd import synthetic
from gdmix_amd.solver import REDeviceSolver, SolverOptions

s = REDeviceSolver(0)
s.set_timing(True)
o = SolverOptions(regularize_bias=False)
for label, b in (("small x200", synthetic.make_batch(200, 16, 4, 1024, seed=5)),
                 ("n=100k k=8 D=65536", synthetic.make_batch(1, 100000, 8, 65536, seed=6, size_dist="const")),
                 ("n=300k k=8 D=65536", synthetic.make_batch(1, 300000, 8, 65536, seed=7, size_dist="const")),
                 ("n=20k k=8 D=65536", synthetic.make_batch(4, 20000, 8, 65536, seed=8, size_dist="const")),
                 ("n=300k k=8 D=64", synthetic.make_batch(1, 300000, 8, 64, seed=9, size_dist="const"))):
    packed = s.pack(b)
    for giant, team in ((1, 0), (0, 1), (0, 0)):
        s.set_giant_nnz(giant)
        s.set_team_nnz(team)
        r = s.solve(packed, o)
        torch.cuda.synchronize()
        ms = np.array(s.last_solve_ms())
        h = r.to_host()
        which = "grid " if giant else ("8team" if team else "block")
        print(f"{label:22s} {which} {ms.sum():9.3f} ms  nit {h['nit'].sum():5d} nfev {h['nfev'].sum():5d}  "
              f"us/eval {1e3 * ms.sum() / h['nfev'].sum():8.1f}  status {np.unique(h['status'])}")
