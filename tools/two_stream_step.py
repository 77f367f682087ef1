"""C2 resident in HBM, the step (gdmix_re_pack + gdmix_re_solve) on ONE stream as bench.py times it, against the same steps dealt over
W contexts with a stream each (W host threads): the pack of one batch next to the solve of another.

    PYTHONPATH=. python tools/two_stream_step.py [entities] [steps] [workers ...]

Every context has its own copy of the batch (what a driver holding W partitions in HBM has)."""
import sys
import threading
import time

import torch

from gdmix_amd import synthetic
from gdmix_amd.solver import REDeviceSolver, SolverOptions

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 24
WORKERS = [int(x) for x in sys.argv[3:]] or [1, 2, 3]
opts = SolverOptions(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
batch = synthetic.make_survey_batch(E, 16, 4, 1024, seed=synthetic.C2_SEED)


class Worker:
    def __init__(self):
        self.solver = REDeviceSolver(0)
        self.stream = torch.cuda.Stream(device=self.solver.device)
        with torch.cuda.stream(self.stream):
            self.raw = self.solver.upload(batch)
            self.stream.synchronize()
        self.res = None

    def steps(self, n):
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                pk = self.solver.pack(self.raw)
                self.res = self.solver.solve(pk, opts)
            self.stream.synchronize()


def run(ws, steps):
    start = threading.Barrier(len(ws) + 1)

    def body(w, n):
        start.wait()
        w.steps(n)
    share = [steps // len(ws) + (i < steps % len(ws)) for i in range(len(ws))]
    th = [threading.Thread(target=body, args=(w, n)) for w, n in zip(ws, share)]
    for t in th:
        t.start()
    torch.cuda.synchronize()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


ws = [Worker() for _ in range(max(WORKERS))]
for w in ws:
    w.steps(2)
ref = None
for nw in WORKERS:
    run(ws[:nw], nw * 2)
    ms = sorted(run(ws[:nw], STEPS) for _ in range(3))
    st = ws[0].res.status
    conv = int(((st >= 0) & (st <= 2)).sum())
    ref = ref or ms[1]
    print(f"{nw} context(s): {ms[1]:.3f} ms per step [{ms[0]:.3f}, {ms[2]:.3f}] = {conv / ms[1] / 1e3:.1f} M entities/s  ({ref / ms[1]:.3f} x one stream)")
for w in ws:
    w.solver.close()
