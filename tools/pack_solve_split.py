"""C2 (1 M entities): gdmix_re_pack and gdmix_re_solve timed apart (HIP events on the launch stream, 10 repetitions each), and the
kernels of one pack by rocprofv3 when run under it.   PYTHONPATH=. python tools/pack_solve_split.py [entities]"""
import sys
import torch
from gdmix_amd import synthetic
from gdmix_amd.solver import REDeviceSolver, SolverOptions

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
s = REDeviceSolver(0)
raw = synthetic.make_device_batch(s.device, E, 16, 4, 1024, seed=20240601) if hasattr(synthetic, "make_device_batch") else s.upload(synthetic.make_batch(E, 16, 4, 1024, seed=1))
opts = SolverOptions(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
for _ in range(3):
    packed = s.pack(raw)
    res = s.solve(packed, opts)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
pm, sm, tm = [], [], []
for _ in range(10):
    ev[0].record()
    packed = s.pack(raw)
    ev[1].record()
    res = s.solve(packed, opts)
    ev[2].record()
    torch.cuda.synchronize()
    pm.append(ev[0].elapsed_time(ev[1])); sm.append(ev[1].elapsed_time(ev[2])); tm.append(ev[0].elapsed_time(ev[2]))
med = lambda a: sorted(a)[len(a) // 2]
print(f"pack {med(pm):.3f} ms  solve {med(sm):.3f} ms  step {med(tm):.3f} ms  (medians of 10)")
