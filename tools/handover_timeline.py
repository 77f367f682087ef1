import sys, time, threading, numpy as np, torch
sys.path.insert(0, ".")
from gdmix_amd import synthetic
from gdmix_amd.solver import REDeviceSolver, SolverOptions
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
batch = synthetic.make_batch(1_000_000, 16, 4, 1024, seed=synthetic.C2_SEED, with_uid=False)
opts = SolverOptions(l2=1.0, regularize_bias=False)
wire = batch.to_wire()
keys = [k for k in REDeviceSolver.WIRE_ARRAYS if wire[k] is not None]
s0 = REDeviceSolver(0); P = s0.pack(batch).P
class Worker:
    def __init__(self):
        self.solver = REDeviceSolver(0); self.stream = torch.cuda.Stream()
        self.wire = dict(wire)
        for k in keys: self.wire[k] = torch.from_numpy(wire[k]).pin_memory()
        self.theta_host = torch.empty(P, dtype=torch.float64).pin_memory()
        self.log = []
    def one(self, t0, tag):
        s = self.solver
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        a = time.perf_counter()
        with torch.cuda.stream(self.stream):
            ev[0].record()
            wd = s.upload_wire(self.wire)
            ev[1].record()
            rd = s.widen(wd); pk = s.pack(rd); res = s.solve(pk, opts)
            ev[2].record()
            self.theta_host[:pk.P].copy_(res.theta_thr, non_blocking=True)
            ev[3].record()
            self.stream.synchronize()
        b = time.perf_counter()
        self.log.append((tag, a - t0, b - t0, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])))
ws = [Worker() for _ in range(W)]
for w in ws: w.one(time.perf_counter(), -1); w.log.clear()
torch.cuda.synchronize()
start = threading.Barrier(W + 1)
t0 = [0.0]
def run(i):
    start.wait()
    for k in range(i, K, W): ws[i].one(t0[0], k)
ths = [threading.Thread(target=run, args=(i,)) for i in range(W)]
for t in ths: t.start()
t0[0] = time.perf_counter(); start.wait()
for t in ths: t.join()
dt = time.perf_counter() - t0[0]
print("workers", W, "partitions", K, "ms/partition %.2f" % (dt / K * 1e3), "ent/s %.1fM" % (K * 1e6 / dt / 1e6))
for i, w in enumerate(ws):
    for (tag, a, b, h2d, comp, d2h) in w.log:
        print("w%d part %2d host %.1f..%.1f ms | h2d %.1f comp %.1f d2h %.1f" % (i, tag, a * 1e3, b * 1e3, h2d, comp, d2h))
