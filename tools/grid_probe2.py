"""Per-phase timing of the team kernels (library built with -DGDMIX_TEAM_PROFILE prints from the device)."""
import torch
from gdmix_amd import synthetic
from gdmix_amd.solver import REDeviceSolver, SolverOptions

s = REDeviceSolver(0)
s.set_timing(True)
o = SolverOptions(regularize_bias=False)
for label, b in (("n=100k k=8 D=65536", synthetic.make_batch(1, 100000, 8, 65536, seed=6, size_dist="const")),
                 ("n=20k k=8 D=65536", synthetic.make_batch(1, 20000, 8, 65536, seed=8, size_dist="const")),
                 ("n=4k k=8 D=65536", synthetic.make_batch(1, 4000, 8, 65536, seed=8, size_dist="const")),
                 ("n=300k k=8 D=64", synthetic.make_batch(1, 300000, 8, 64, seed=9, size_dist="const"))):
    packed = s.pack(b)
    for giant, team in ((1, 0), (0, 1), (0, 0)):
        s.set_giant_nnz(giant)
        s.set_team_nnz(team)
        print(label, "grid" if giant else ("team" if team else "block"), flush=True)
        r = s.solve(packed, o)
        torch.cuda.synchronize()
