"""Randomised parity sweep: the HIP path (through the C ABI) against the CPU oracle on seeded random batches — shapes,
solver options, warm starts and kernel routing drawn at random. Not a test (tests/ holds the fixed cases); a tool to
look for rare disagreements on the GPU box.

    PYTHONPATH=.:tests python tools/fuzz_parity.py [cases] [first_seed]
"""
import sys
import time

import numpy as np

from gdmix_amd import synthetic
from gdmix_amd.solver import REDeviceSolver, SolverOptions
from oracle import oracle

sys.path.insert(0, "tests")
from helpers import per_entity_rel_err, well_posed_mask  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
solver = REDeviceSolver(0)
bad = 0
t0 = time.time()
worst = 0.0
for case in range(cases):
    rng = np.random.default_rng(seed0 + case)
    shape = rng.choice(["c2", "ragged", "zipf", "ml", "ml20m", "wide", "tall", "tiny"])
    if shape == "c2":
        b = synthetic.make_batch(int(rng.integers(50, 3000)), int(rng.integers(2, 40)), int(rng.choice([1, 2, 4, 8])), int(rng.choice([64, 1024, 65536])),
                                 seed=seed0 + case, with_uid=False)
    elif shape == "ragged":
        b = synthetic.make_ragged_batch(int(rng.integers(20, 1500)), seed=seed0 + case, D=int(rng.choice([30, 200, 5000])),
                                        max_n=int(rng.integers(2, 120)), max_k=int(rng.integers(1, 20)))
    elif shape == "zipf":
        b = synthetic.make_batch(int(rng.integers(200, 4000)), 32, 8, int(rng.choice([4096, 65536])), seed=seed0 + case, size_dist="zipf",
                                 with_uid=False)
    elif shape == "ml":
        b = synthetic.make_movielens_like(int(rng.integers(50, 1500)), str(rng.choice(["per_user", "per_movie"])), seed=seed0 + case)
    elif shape == "ml20m":   # MovieLens-20M entity sizes (tall and skinny: the tall kernel, the counting pack path)
        b = synthetic.make_movielens_20m(str(rng.choice(["per_user", "per_movie"])), seed=seed0 + case, entities=int(rng.integers(20, 400)))
    elif shape == "wide":    # few samples, many features
        b = synthetic.make_batch(int(rng.integers(3, 40)), int(rng.integers(2, 30)), int(rng.choice([64, 128, 256])), 65536, seed=seed0 + case,
                                 size_dist="const", with_uid=False)
    elif shape == "tall":    # many samples, few features
        b = synthetic.make_batch(int(rng.integers(1, 6)), int(rng.integers(3000, 60000)), int(rng.choice([1, 2, 4])), int(rng.choice([8, 64, 512])),
                                 seed=seed0 + case, size_dist="const", with_uid=False)
    else:
        b = synthetic.make_batch(int(rng.integers(1, 300)), 1, int(rng.choice([1, 2, 4])), 16, seed=seed0 + case, size_dist="const", with_uid=False)
    has_intercept = bool(rng.random() < 0.8)
    kw = dict(l2=float(rng.choice([0.01, 0.1, 1.0, 10.0])), regularize_bias=bool(rng.random() < 0.5) and has_intercept, has_intercept=has_intercept,
              m=int(rng.choice([1, 3, 10])), max_iter=int(rng.choice([2, 15, 100])), ftol=float(rng.choice([1e-12, 1e-7])),
              variance_mode=int(rng.choice([0, 0, 1])))
    opts_j = dict(l2=kw["l2"], regularize_bias=kw["regularize_bias"], has_intercept=has_intercept)
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    if shape in ("c2", "ragged", "ml", "tiny") and np.diff(pk["ent_feat_ptr"]).max() < 300 and rng.random() < 0.3:
        kw["variance_mode"] = 2     # FULL: a dense p x p inverse per entity
    packed = solver.pack(b, has_intercept=has_intercept)
    th0 = None
    if rng.random() < 0.3:
        th0 = 0.1 * rng.standard_normal(int(packed.P))
    routing = dict(giant=int(rng.choice([16777216, 16777216, 200000, 1])), team=int(rng.choice([16384, 16384, 2048, 256])),
                   mask=int(rng.choice([7, 7, 1])), tall=int(rng.choice([32, 32, 1, 0])))
    solver.set_giant_nnz(routing["giant"]); solver.set_team_nnz(routing["team"]); solver.set_kernel_mask(routing["mask"])
    solver.set_tall_min_n(routing["tall"])
    try:
        res = solver.solve(packed, SolverOptions(**kw), theta0=th0).to_host()
    finally:
        solver.set_giant_nnz(16777216); solver.set_team_nnz(16384); solver.set_kernel_mask(7); solver.set_tall_min_n(solver.TALL_MIN_N_DEFAULT)
    ref = oracle.solve(pk, b.val, b.y, b.offset, b.weight, oracle.make_opts(**kw), theta0=th0)
    coef_ptr = packed.coef_ptr_host()
    wp = well_posed_mask(b, opts_j)
    err = per_entity_rel_err(res["theta"], ref["theta"], coef_ptr)
    # Which entities can be compared iteration for iteration? Long runs on badly conditioned entities (small l2, m = 1)
    # amplify rounding: the oracle run again from a start moved by ~1 ulp already takes another number of iterations and
    # stops up to 1e-3 away. Those entities are compared against that sensitivity, the stable ones strictly.
    sens = np.zeros(b.E)
    stable = np.ones(b.E, bool)
    for j, mag in enumerate((1e-15, 1e-14, 1e-13)):   # the divergence is chaotic: a few samples of it, up to the size of the
        # difference between two summation orders over 1e4 .. 1e5 samples (which is what the device and the oracle differ by)
        jig = mag * np.random.default_rng(j + 1).standard_normal(int(packed.P))
        pert = oracle.solve(pk, b.val, b.y, b.offset, b.weight, oracle.make_opts(**kw), theta0=jig if th0 is None else th0 * (1.0 + jig))
        sj = per_entity_rel_err(pert["theta"], ref["theta"], coef_ptr)
        sens = np.maximum(sens, sj)
        stable &= (pert["status"] == ref["status"]) & (pert["nit"] == ref["nit"]) & (sj < 1e-9)
    wp_all = wp
    unstable_bad = wp & ~stable & (err > 1e-6) & (err > 1000.0 * np.maximum(sens, 1e-12))
    wp = wp & stable
    # a FACTR stop decides on differences at rounding level: iteration counts may differ there, the result may not
    same = (res["status"] == ref["status"]) & (res["nit"] == ref["nit"])
    strict = wp & (ref["status"] != 1) & (res["status"] != 1)
    problems = []
    if not np.array_equal(packed.unique_global().cpu().numpy(), pk["unique_global"]):
        problems.append("pack: unique_global differs")
    if np.any(res["status"] < 0) or np.any(res["status"] > 4):
        problems.append(f"status out of range: {np.unique(res['status'])}")
    if strict.any() and not same[strict].all():
        k = np.flatnonzero(strict & ~same)
        problems.append(f"{k.size} entities differ in status/nit, e.g. {k[:3]}: dev {res['status'][k[:3]]}/{res['nit'][k[:3]]} "
                        f"ref {ref['status'][k[:3]]}/{ref['nit'][k[:3]]}")
    if strict.any() and not np.array_equal(res["nfev"][strict & same], ref["nfev"][strict & same]):   # scipy's funcalls, counted on the device
        k = np.flatnonzero(strict & same & (res["nfev"] != ref["nfev"]))
        problems.append(f"{k.size} entities differ in nfev, e.g. {k[:3]}: dev {res['nfev'][k[:3]]} ref {ref['nfev'][k[:3]]}")
    # a FACTR stop with a loose ftol leaves the coefficients determined to about sqrt(ftol) only
    tol = np.where((res["status"] == 1) | (ref["status"] == 1), 1e-6 if kw["ftol"] <= 1e-12 else 1e-3, 1e-6)
    if wp.any() and np.any(err[wp] > tol[wp]):
        k = int(np.argmax(np.where(wp, err / tol, 0)))
        problems.append(f"theta rel err {err[k]:.2e} at entity {k} (n={b.ent_n()[k]}, nnz={b.ent_nnz()[k]}, status {res['status'][k]}/{ref['status'][k]})")
    if kw["variance_mode"] in (1, 2) and wp.any():
        v, vr = res["variance"], ref["variance"]
        m = np.zeros(coef_ptr[-1], bool)
        for e in np.flatnonzero(wp & same):
            m[coef_ptr[e]:coef_ptr[e + 1]] = True
        # FULL inverts the Hessian (Cholesky on the device, LU in the oracle): the agreement is limited by its conditioning
        if m.any() and not np.allclose(v[m], vr[m], rtol=1e-6 if kw["variance_mode"] == 1 else 1e-4):
            k = int(np.argmax(np.where(m, np.abs(v - vr) / np.maximum(np.abs(vr), 1e-300), 0)))
            problems.append(f"variance differs (mode {kw['variance_mode']}): {v[k]:.6e} vs {vr[k]:.6e}")
    if unstable_bad.any():
        k = np.flatnonzero(unstable_bad)
        problems.append(f"{k.size} rounding-sensitive entities are further off than the oracle's own sensitivity explains, e.g. {k[:3]}: "
                        f"err {err[k[:3]]} sensitivity {sens[k[:3]]}")
    # scoring pass with the oracle's coefficients and a random set of entities without a model
    hm = (rng.random(b.E) < 0.85).astype(np.uint8) if rng.random() < 0.5 else None
    lo_d, pc_d = solver.score(packed, ref["theta"], hm)
    lo_o, pc_o = oracle.score(pk, b.val, b.offset, ref["theta"], has_intercept, hm)
    fin = np.isfinite(lo_o)
    if not np.allclose(lo_d.cpu().numpy()[fin], lo_o[fin], rtol=3e-6, atol=3e-6) or not np.allclose(pc_d.cpu().numpy()[fin], pc_o[fin], rtol=3e-5, atol=3e-5):
        problems.append("scores differ")
    worst = max(worst, float(err[wp].max()) if wp.any() else 0.0)
    tag = "ok " if not problems else "BAD"
    if problems:
        bad += 1
    if problems or case % 10 == 0:
        print(f"{tag} case {seed0 + case} {shape:6s} E={b.E} N={b.N} Z={b.Z} stable {int(wp.sum())}/{int(wp_all.sum())} {kw} warm={th0 is not None} {routing}"
              + ("".join("\n      " + p for p in problems)), flush=True)
print(f"{cases} cases, {bad} with disagreements, worst well-posed theta rel err {worst:.2e}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
