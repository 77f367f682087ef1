#!/usr/bin/env python3
"""Golden vectors for the L-BFGS-B exits and line-search branches the first fixture set never reaches
(VERDICT r1 item 4): FACTR stops (`lbfgs_tolerance` 1e-7 / 1e-4 through binary_logistic_regression.py:223-231),
long line searches (> 10 evaluations in one search), the `maxls` abort / ABNORMAL_TERMINATION_IN_LNSRCH and
skipped curvature pairs — whatever the REFERENCE ITSELF produces on seeded, badly scaled entities.

Same rules as generate_fixtures.py (which this script imports for the reference plumbing): runs only in the
build container, drives prepare_jobs -> TrainingJobConsumer -> fit of /root/reference, writes data only.

Parity class of an entity in these sets. Long runs on ill-conditioned entities amplify rounding until the
reference's own answer is not reproducible by anything: the reference is therefore run four more
times — from a start moved by +-1e-14, with every sample weight moved by +-1 ulp, and with the weights moved by
-2 .. 2 ulps sample by sample and l2 by +-1 ulp (noise of the size another summation order, BLAS build or FMA
contraction inside scipy would inject). Entities whose eleven runs agree (same
nit / nfev / stop reason, theta within 1e-9) are `exp_strict` = 1 and compared iteration for iteration; the others
are compared against invariants only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_exit_fixtures.py
"""
import os
import queue
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import generate_fixtures as gf  # noqa: E402  (stubs tensorflow / fastavro, imports the reference)

from gdmix_amd import synthetic  # noqa: E402
from gdmix_amd.batch import RawBatch, concat  # noqa: E402
from oracle import oracle  # noqa: E402  (candidate DETECTOR only: every expected value below comes from the reference)


def run_reference_with_spread(b, num_features, jiggle=1e-14, **opts):
    """run_reference + for every entity two more fit() calls from theta0 +- jiggle -> strict mask."""
    lr = gf.BinaryLogisticRegressionTrainer(regularize_bias=opts["regularize_bias"], lambda_l2=opts["l2"],
                                            precision=opts["tol"] / np.finfo(float).eps,
                                            num_lbfgs_corrections=opts["m"], max_iter=opts["max_iter"],
                                            has_intercept=opts["has_intercept"])
    raw, strict = [], []
    orig_fit = lr.fit

    def fit(**kw):
        res = orig_fit(**kw)
        raw.append((res, kw["theta_initial"].copy()))
        ok = True
        # (a) the start moved by +-jiggle; (b) every sample weight moved by one or two ulps — which perturbs f and g at the
        # 1e-16 relative level everywhere, like another summation order / BLAS build / FMA contraction would
        trials = [dict(kw, theta_initial=kw["theta_initial"] + sign * jiggle) for sign in (1.0, -1.0)]
        w64 = np.asarray(kw["weights"], np.float64)
        trials += [dict(kw, weights=w64 * (1.0 + 2.0 ** -52)), dict(kw, weights=w64 * (1.0 - 2.0 ** -52))]
        nrng = np.random.default_rng(len(raw))   # ... and by -2 .. 2 ulps sample by sample (what a summation order does to a sum)
        trials += [dict(kw, weights=w64 * (1.0 + nrng.integers(-2, 3, w64.size) * 2.0 ** -52)) for _ in range(6)]
        lam = lr.lambda_l2
        for ti, kw2 in enumerate(trials):
            # ... and the regulariser by an ulp (its sum is placed differently in every implementation)
            lr.lambda_l2 = lam * (1.0 + (2.0 ** -52 if ti % 2 else -2.0 ** -52)) if ti >= 4 else lam
            r2 = orig_fit(**kw2)
            lr.lambda_l2 = lam
            same = (r2[0][2]["nit"] == res[0][2]["nit"] and r2[0][2]["funcalls"] == res[0][2]["funcalls"]
                    and gf.status_code(r2[0][2]["task"]) == gf.status_code(res[0][2]["task"]))
            den = max(np.max(np.abs(res[0][0])), 1e-300)
            ok = ok and same and np.max(np.abs(r2[0][0] - res[0][0])) / den <= 1e-9
        strict.append(ok)
        return res
    lr.fit = fit
    q = queue.Queue()
    consumer = gf.jc.TrainingJobConsumer(lr, "fixture", q, enable_local_indexing=True, sparsity_threshold=1e-4,
                                         variance_mode=None)
    model_params = SimpleNamespace(partition_entity="entity", feature_bag="bag", offset_column_name="offset")
    schema = SimpleNamespace(uid_column_name="uid", label_column_name="label", weight_column_name="weight")
    out = []
    for jid in gf.jc.prepare_jobs(lambda: gf.tf_batches(b, 16), model_params, schema, num_features, {}, True, q,
                                  opts["has_intercept"]):
        out.append(consumer(jid))
    return out, raw, np.array(strict, bool)


def case(name, b, num_features, keep=None, **kw):
    opts = dict(l2=1.0, regularize_bias=True, has_intercept=True, m=10, max_iter=100, tol=1e-12, variance_mode=None)
    opts.update(kw)
    out, raw, strict = run_reference_with_spread(b, num_features, **opts)
    res = gf.pack_results(b, out, raw, opts["has_intercept"], True)
    if keep is not None:   # keep the interesting entities only
        sel = np.flatnonzero(keep(res, strict))
        if sel.size == 0:
            print(f"{name}: nothing selected")
            return None
        b = b.select(sel)
        out, raw, strict = run_reference_with_spread(b, num_features, **opts)
        res = gf.pack_results(b, out, raw, opts["has_intercept"], True)
    opts["num_features"] = num_features
    opts["local"] = True
    res["strict"] = strict.astype(np.uint8)
    gf.save(name, b, opts, res)
    extra = res["nfev"] - res["nit"] - 1
    print(f"   strict {int(strict.sum())}/{strict.size}; nfev-nit-1 >= 8: {int((extra >= 8).sum())}; "
          f"status histogram {np.bincount(res['status'], minlength=5).tolist()}")
    return res


def extreme_entity(rng, eid):
    """1-4 samples, 1-2 of 4 features, values 10^U(-4,4), offsets up to +-1e32 on some or all samples: saturated
    sigmoids make the objective piecewise linear over many decades, which is what drives L-BFGS-B off its common path."""
    n = int(rng.integers(1, 5))
    k = int(rng.integers(1, 3))
    cols = np.stack([rng.choice(4, k, replace=False) for _ in range(n)]).reshape(-1)
    vals = (rng.standard_normal(n * k) * 10.0 ** rng.uniform(-4, 4)).astype(np.float32)
    off = (rng.standard_normal(n) * 10.0 ** rng.uniform(0, 32)).astype(np.float32)
    if rng.random() < 0.5:
        m = rng.random(n) < 0.5
        off[m] = rng.standard_normal(int(m.sum())).astype(np.float32)
    y = (rng.random(n) < 0.5).astype(np.float32)
    return RawBatch(ent_row_ptr=[0, n], row_nnz_ptr=np.arange(n + 1) * k, col_global=cols, val=vals, y=y, offset=off,
                    uid=np.arange(n), entity_ids=[str(eid)])


def extreme_cases():
    """Search seeded extreme entities for the branches no ordinary entity takes — the curvature-skip rule, the g'd >= 0
    restart, the maxls abort (with and without history) and ABNORMAL_TERMINATION_IN_LNSRCH — using the oracle's branch
    counters to spot candidates, then record what the REFERENCE does on them (and whether it reproduces itself)."""
    quota = {"skipped_pairs": 12, "gd_restarts": 6, "maxls_aborts": 16, "abnormal": 12, "long_search": 10}
    for ci, (l2, rb) in enumerate([(0.0, False), (1e-6, True), (1e-3, True), (0.0, True)]):
        rng = np.random.default_rng(7000 + ci)
        got = {k: 0 for k in quota}
        picked = []
        o = oracle.make_opts(l2=l2, regularize_bias=rb)
        for it in range(150000):
            b = extreme_entity(rng, len(picked))
            pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
            oracle.branch_counts()
            r = oracle.solve(pk, b.val, b.y, b.offset, None, o)
            bc = oracle.branch_counts()
            tags = [k for k in ("skipped_pairs", "gd_restarts", "maxls_aborts") if bc[k]]
            if r["status"][0] == 4:
                tags.append("abnormal")
            if bc["max_evals_in_one_search"] > 10 and not bc["maxls_aborts"]:
                tags.append("long_search")
            want = [t for t in tags if got[t] < quota[t]]
            if want:
                for t in tags:
                    got[t] += 1
                picked.append(b)
            if all(got[k] >= quota[k] for k in quota):
                break
        print(f"extreme l2={l2:g} rb={rb}: picked {len(picked)} entities after {it + 1} candidates, oracle tags {got}")
        if picked:
            case(f"exit_extreme_{ci:02d}", concat(picked), 4, l2=l2, regularize_bias=rb)


def main():
    extreme_cases()
    c2 = synthetic.make_batch(300, 16, 4, 1024, seed=synthetic.C2_SEED)
    # (f) FACTR stops from the reference itself: lbfgs_tolerance above the default 1e-12
    case("exit_factr_1e-7", c2.select(np.arange(150)), 1024, tol=1e-7, regularize_bias=False)
    case("exit_factr_1e-4", c2.select(np.arange(150, 300)), 1024, tol=1e-4, regularize_bias=True)
    case("exit_factr_1e-4_m3_weights", synthetic.make_batch(100, 16, 4, 1024, seed=11, random_weights=True), 1024, tol=1e-4, m=3,
         regularize_bias=False)
    # (g) badly scaled entities: values x1e3, offsets x50, l2 = 0 and tiny; keep what leaves the common path
    hist = np.zeros(5, np.int64)
    total = 0
    picks = []
    for seed, (vs, lo, l2, mean_n, k, dist) in enumerate([
            (1e3, 50.0, 0.0, 6, 3, "poisson"), (1e3, 50.0, 1e-6, 6, 3, "poisson"), (30.0, 50.0, 0.0, 12, 4, "poisson"),
            (1e3, 5.0, 1e-3, 4, 2, "geometric"), (300.0, 50.0, 1e-9, 3, 2, "geometric"), (1e3, 50.0, 0.0, 24, 4, "poisson"),
            (1e2, 20.0, 1e-4, 8, 3, "poisson"), (1e4, 100.0, 0.0, 5, 2, "geometric")]):
        b = synthetic.make_batch(250, mean_n, k, 64 * k, seed=900 + seed, size_dist=dist, l_offset=lo, value_scale=vs)
        for rb in (False, True):
            opts = dict(l2=l2, regularize_bias=rb, has_intercept=True, m=10, max_iter=100, tol=1e-12)
            out, raw, strict = run_reference_with_spread(b, 64 * k, **opts)
            res = gf.pack_results(b, out, raw, True, True)
            hist += np.bincount(res["status"], minlength=5)
            total += b.E
            extra = res["nfev"] - res["nit"] - 1
            interesting = (res["status"] == 4) | (res["status"] == 1) | (extra >= 8)
            print(f"search vs={vs:g} lo={lo:g} l2={l2:g} n~{mean_n} rb={rb}: status {np.bincount(res['status'], minlength=5).tolist()} "
                  f"max extra evals {int(extra.max())} strict {int(strict.sum())}/{b.E} interesting {int(interesting.sum())} "
                  f"(strict among them {int((interesting & strict).sum())})")
            picks.append((b, 64 * k, opts, interesting, strict))
    print(f"searched {total} reference solves: status histogram {hist.tolist()}")
    # one fixture per option set that produced interesting entities, at most 40 entities each (strict ones first)
    n_case = 0
    for b, D, opts, interesting, strict in picks:
        if not interesting.any():
            continue
        idx = np.flatnonzero(interesting)
        idx = np.concatenate([idx[strict[idx]], idx[~strict[idx]]])[:40]
        idx.sort()
        name = f"exit_hard_{n_case:02d}"
        case(name, b.select(idx), D, **opts)
        n_case += 1


if __name__ == "__main__":
    main()
