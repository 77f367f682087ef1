"""One case of tools/fuzz_parity.py in detail: per entity status / nit / nfev / f / |g| of the device (under several routings)
and of the oracle.   PYTHONPATH=.:tests python tools/fuzz_case_detail.py <seed> [entity ...]"""
import sys

import numpy as np

sys.path.insert(0, "tests")
sys.argv_backup = list(sys.argv)
seed = int(sys.argv[1])
ents = [int(x) for x in sys.argv[2:]]
from gdmix_amd import synthetic  # noqa: E402
from gdmix_amd.solver import REDeviceSolver, SolverOptions  # noqa: E402
from helpers import per_entity_rel_err  # noqa: E402
from oracle import oracle  # noqa: E402

rng = np.random.default_rng(seed)
shape = rng.choice(["c2", "ragged", "zipf", "ml", "ml20m", "wide", "tall", "tiny"])   # as fuzz_parity.py draws (round 3 list)
if shape == "c2":
    b = synthetic.make_batch(int(rng.integers(50, 3000)), int(rng.integers(2, 40)), int(rng.choice([1, 2, 4, 8])), int(rng.choice([64, 1024, 65536])), seed=seed, with_uid=False)
elif shape == "ragged":
    b = synthetic.make_ragged_batch(int(rng.integers(20, 1500)), seed=seed, D=int(rng.choice([30, 200, 5000])), max_n=int(rng.integers(2, 120)), max_k=int(rng.integers(1, 20)))
elif shape == "zipf":
    b = synthetic.make_batch(int(rng.integers(200, 4000)), 32, 8, int(rng.choice([4096, 65536])), seed=seed, size_dist="zipf", with_uid=False)
elif shape == "ml":
    b = synthetic.make_movielens_like(int(rng.integers(50, 1500)), str(rng.choice(["per_user", "per_movie"])), seed=seed)
elif shape == "ml20m":
    b = synthetic.make_movielens_20m(str(rng.choice(["per_user", "per_movie"])), seed=seed, entities=int(rng.integers(20, 400)))
elif shape == "wide":
    b = synthetic.make_batch(int(rng.integers(3, 40)), int(rng.integers(2, 30)), int(rng.choice([64, 128, 256])), 65536, seed=seed, size_dist="const", with_uid=False)
elif shape == "tall":
    b = synthetic.make_batch(int(rng.integers(1, 6)), int(rng.integers(3000, 60000)), int(rng.choice([1, 2, 4])), int(rng.choice([8, 64, 512])), seed=seed, size_dist="const", with_uid=False)
else:
    b = synthetic.make_batch(int(rng.integers(1, 300)), 1, int(rng.choice([1, 2, 4])), 16, seed=seed, size_dist="const", with_uid=False)
has_intercept = bool(rng.random() < 0.8)
kw = dict(l2=float(rng.choice([0.01, 0.1, 1.0, 10.0])), regularize_bias=bool(rng.random() < 0.5) and has_intercept, has_intercept=has_intercept,
          m=int(rng.choice([1, 3, 10])), max_iter=int(rng.choice([2, 15, 100])), ftol=float(rng.choice([1e-12, 1e-7])), variance_mode=int(rng.choice([0, 0, 1])))
solver = REDeviceSolver(0)
pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
if shape in ("c2", "ragged", "ml", "tiny") and np.diff(pk["ent_feat_ptr"]).max() < 300 and rng.random() < 0.3:   # as fuzz_parity.py draws
    kw["variance_mode"] = 2
packed = solver.pack(b, has_intercept=has_intercept)
th0 = 0.1 * rng.standard_normal(int(packed.P)) if rng.random() < 0.3 else None
print(shape, "E", b.E, kw, "warm", th0 is not None)
ref = oracle.solve(pk, b.val, b.y, b.offset, b.weight, oracle.make_opts(**kw), theta0=th0)
cp = packed.coef_ptr_host()
ents = ents or list(range(min(b.E, 8)))
# the oracle's own spread: the same solve from starts moved by 1e-15 .. 1e-13 (fuzz_parity.py)
for j, mag in enumerate((1e-15, 1e-14, 1e-13)):
    jig = mag * np.random.default_rng(j + 1).standard_normal(int(packed.P))
    pert = oracle.solve(pk, b.val, b.y, b.offset, b.weight, oracle.make_opts(**kw), theta0=jig if th0 is None else th0 * (1.0 + jig))
    sj = per_entity_rel_err(pert["theta"], ref["theta"], packed.coef_ptr_host())
    for e in ents:
        print(f"oracle start moved by {mag:g}: entity {e}: status {pert['status'][e]} nit {pert['nit'][e]} nfev {pert['nfev'][e]} f {pert['fval'][e]:.15g} "
              f"|g| {pert['gnorm'][e]:.3e}  theta rel err {sj[e]:.3e}")
for e in ents:
    print(f"entity {e}: n={b.ent_n()[e]} nnz={b.ent_nnz()[e]} p={cp[e + 1] - cp[e]}   oracle status {ref['status'][e]} nit {ref['nit'][e]} nfev {ref['nfev'][e]} "
          f"f {ref['fval'][e]:.15g} |g| {ref['gnorm'][e]:.3e}")
for label, giant, team, mask, tall in (("default", 16777216, 16384, 7, 32), ("no tall kernel", 16777216, 16384, 7, 0), ("device-wide", 1, 16384, 7, 0),
                                      ("teams from 256", 16777216, 256, 7, 0), ("teams from 256, wave kernels, tall", 16777216, 256, 1, 32),
                                      ("register wave only", 16777216, 16384, 1, 0)):
    solver.set_giant_nnz(giant); solver.set_team_nnz(team); solver.set_kernel_mask(mask); solver.set_tall_min_n(tall)
    res = solver.solve(packed, SolverOptions(**kw), theta0=th0).to_host()
    err = per_entity_rel_err(res["theta"], ref["theta"], cp)
    classes = {n: c for n, c in solver.class_counts(packed) if c}
    print(label, classes)
    for e in ents:
        print(f"   entity {e}: status {res['status'][e]} nit {res['nit'][e]} nfev {res['nfev'][e]} f {res['fval'][e]:.15g} |g| {res['gnorm'][e]:.3e}  theta rel err {err[e]:.3e}")
