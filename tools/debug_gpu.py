"""Staged device check with a sync after every stage (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gdmix_amd import synthetic
from gdmix_amd.solver import REDeviceSolver, SolverOptions
from oracle import oracle

E = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b = synthetic.make_batch(E, 16, 4, 1024, seed=3)
s = REDeviceSolver(0)
print("ctx ok", flush=True)
packed = s.pack(b)
torch.cuda.synchronize()
print("pack ok D", packed.D, "max", packed.max_p, packed.max_n, packed.max_nnz, flush=True)
pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
print("D oracle", pk["D"])
for name in ("ent_feat_ptr", "ent_nnz_ptr", "unique_global", "csr_col", "row_ptr"):
    got = getattr(packed, name)().cpu().numpy()
    print(name, "equal:", np.array_equal(got, pk[name]), flush=True)
    if not np.array_equal(got, pk[name]):
        print(" got", got[:40], "\n exp", pk[name][:40])
kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
lim = int(os.environ.get("LDS_LIMIT", "65536"))
s.set_wave_lds_limit(lim)
res = s.solve(packed, SolverOptions(**kw))
torch.cuda.synchronize()
print("solve ok", s.class_counts(packed), flush=True)
r = res.to_host()
ref = oracle.solve(pk, b.val, b.y, b.offset, None, oracle.make_opts(**kw))
print("nit dev", r["nit"][:16], "\nnit ref", ref["nit"][:16])
print("max abs theta diff", np.max(np.abs(r["theta"] - ref["theta"])))
print("status", np.bincount(r["status"], minlength=5))
