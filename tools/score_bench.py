"""Throughput of the scoring pass (gdmix_re_score: logits X theta + offset for every sample of a packed batch).

    PYTHONPATH=. python tools/score_bench.py [c2|zipf|c5mean] [entities]
"""
import sys

import torch

from gdmix_amd import synthetic
from gdmix_amd.solver import REDeviceSolver

shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
if shape == "c2":
    b = synthetic.make_batch(E, 16, 4, 1024, seed=synthetic.C2_SEED, with_uid=False)
elif shape == "zipf":
    b = synthetic.make_batch(E, 32, 8, 65536, seed=synthetic.C5_SEED, size_dist="zipf", with_uid=False)
else:
    b = synthetic.make_batch(E, 32, 8, 65536, seed=synthetic.C5_SEED, with_uid=False)
s = REDeviceSolver(0)
pk = s.pack(s.upload(b))
theta = torch.randn(int(pk.P), dtype=torch.float64, device=s.device)
for _ in range(3):
    logit, per = s.score(pk, theta)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
reps = 20
ev[0].record()
for _ in range(reps):
    logit, per = s.score(pk, theta)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
nbytes = 8.0 * b.Z + 4.0 * b.N + 4.0 * (b.N + b.E) + 8.0 * pk.P + 8.0 * b.N + 24.0 * b.E
print(f"{shape}: {b.E} entities, {b.N} samples, {b.Z} nnz: {ms:.3f} ms per pass, {b.N / ms / 1e3:.1f} M samples/s, "
      f"{nbytes / ms / 1e6:.0f} GB/s of algorithmic bytes ({nbytes / 1e6:.0f} MB), {nbytes / ms / 1e6 / 8000:.1%} of HBM peak")
# check against a host computation on a slice
lo = logit.cpu().numpy()
th = theta.cpu().numpy()
cp = pk.coef_ptr_host()
fp = pk.ent_feat_ptr().cpu().numpy()
uq = pk.unique_global().cpu().numpy()
for e in (0, b.E // 2, b.E - 1):
    r0, r1 = b.ent_row_ptr[e], b.ent_row_ptr[e + 1]
    m = dict(zip(uq[fp[e]:fp[e + 1]].tolist(), th[cp[e] + 1:cp[e + 1]].tolist()))
    for i in range(r0, min(r1, r0 + 3)):
        k0, k1 = b.row_nnz_ptr[i], b.row_nnz_ptr[i + 1]
        z = th[cp[e]] + sum(float(b.val[k]) * m[int(b.col_global[k])] for k in range(k0, k1)) + float(b.offset[i])
        assert abs(z - lo[i]) <= 1e-5 * max(1.0, abs(z)), (e, i, z, lo[i])
print("spot check ok")
