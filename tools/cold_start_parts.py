"""Where the wall time of a cold `python -m gdmix_amd.gdmix` process goes before its first partition is solved: interpreter + imports,
torch, the HIP context, loading libgdmix_re.so and its code objects (one per translation unit, loaded by the runtime at the first
launch of one of its kernels), the first pack / solve (kernel code paged in), a second solve. Exploration tool."""
import os, sys, time
t0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
marks = []
def mark(name):
    marks.append((name, time.perf_counter()))
import numpy as np
mark("numpy")
from gdmix_amd import synthetic
from gdmix_amd.solver import REDeviceSolver, SolverOptions, load_library
mark("gdmix_amd imports (no torch)")
import torch
mark("import torch")
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
mark("HIP context (first tensor)")
load_library()
mark("dlopen libgdmix_re.so")
s = REDeviceSolver(0)
mark("gdmix_re_create")
b = synthetic.make_survey_batch(125_000, 16, 4, 1024, seed=1)
mark("(generate a 125k-entity partition)")
raw = s.upload(b)
torch.cuda.synchronize()
mark("upload")
opts = SolverOptions(l2=1.0, regularize_bias=False)
for k in range(3):
    pk = s.pack(raw)
    torch.cuda.synchronize()
    mark(f"pack #{k + 1}")
    r = s.solve(pk, opts)
    torch.cuda.synchronize()
    mark(f"solve #{k + 1}")
sc = s.score(pk, r.theta_thr)
torch.cuda.synchronize()
mark("score #1")
prev = t0
for name, t in marks:
    print(f"{name:45s} {1e3 * (t - prev):9.1f} ms")
    prev = t
print(f"{'total':45s} {1e3 * (prev - t0):9.1f} ms")
