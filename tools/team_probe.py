"""Per-phase timing of the workgroup-class team kernel under full occupancy (library built with
GDMIX_EXTRA_FLAGS=-DGDMIX_TEAM_PROFILE prints one line per entity from the device)."""
import sys
import torch
from gdmix_amd import synthetic
from gdmix_amd.solver import REDeviceSolver, SolverOptions

n = int(sys.argv[1]) if len(sys.argv) > 1 else 580
E = int(sys.argv[2]) if len(sys.argv) > 2 else 512
s = REDeviceSolver(0)
s.set_timing(True)
o = SolverOptions(regularize_bias=False)
b = synthetic.make_batch(E, n, 8, 65536, seed=6, size_dist="const")
packed = s.pack(b)
r = s.solve(packed, o)
torch.cuda.synchronize()
print("ms", [round(x, 3) for x in s.last_solve_ms() if x > 0], "nfev", r.nfev.double().mean().item())
