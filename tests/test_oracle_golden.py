"""The CPU oracle (oracle/re_oracle.c) against the golden vectors generated from the reference itself
(tests/golden/generate_fixtures.py). This pins the oracle: every later HIP-vs-oracle comparison is
anchored to the reference's own TF/scipy arithmetic through these fixtures."""
import numpy as np
import pytest

from helpers import (REL_TOL_ORACLE, check_d_class, fixture_names, load_fixture, opts_kwargs, parity_mask, per_entity_rel_err,
                     well_posed_mask)
from oracle import oracle


@pytest.mark.parametrize("name", fixture_names())
def test_oracle_matches_reference_fixture(name):
    b, opts, exp, _ = load_fixture(name)
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    # np.unique per entity: bit exact
    assert np.array_equal(pk["unique_global"], exp["unique_global"])
    assert np.array_equal(pk["ent_feat_ptr"], exp["ent_feat_ptr"])
    ic = 1 if opts["has_intercept"] else 0
    coef_ptr = pk["ent_feat_ptr"] + np.arange(b.E + 1) * ic
    o = oracle.make_opts(**opts_kwargs(opts))
    th0 = exp["theta0"] if np.any(exp["theta0"]) else None
    r = oracle.solve(pk, b.val, b.y, b.offset, b.weight, o, theta0=th0)
    wp = parity_mask(b, opts, exp)
    assert wp.any()
    err = per_entity_rel_err(r["theta"], exp["theta"], coef_ptr)
    assert err[wp].max() <= REL_TOL_ORACLE, f"theta rel err {err[wp].max():.3e}"
    # same trajectory: identical iteration and evaluation counts and stop reason on well-posed entities
    assert np.array_equal(r["nit"][wp], exp["nit"][wp])
    assert np.array_equal(r["nfev"][wp], exp["nfev"][wp])
    assert np.array_equal(r["status"][wp], exp["status"][wp])
    # (after ABNORMAL_TERMINATION_IN_LNSRCH L-BFGS-B restores x, g and f of the last iterate; scipy's Python driver keeps the f
    # of the last trial point it evaluated in its own variable and reports that one. Only theta goes into the model.)
    fv = wp & (exp["status"] != 4)
    np.testing.assert_allclose(r["fval"][fv], exp["fval"][fv], rtol=1e-10, atol=1e-14)
    # thresholded coefficients: identical zero pattern
    thr_err = per_entity_rel_err(r["theta_thr"], exp["theta_thr"], coef_ptr)
    assert thr_err[wp].max() <= REL_TOL_ORACLE
    assert np.array_equal((r["theta_thr"] == 0)[_mask(coef_ptr, wp)], (exp["theta_thr"] == 0)[_mask(coef_ptr, wp)])
    if "variance" in exp:
        np.testing.assert_allclose(r["variance"], exp["variance"], rtol=1e-7, atol=0)
    # degenerate class: invariants only — the full contract: gradient, sign of the intercept, thresholded
    # zero pattern, saturated predictions
    dg = ~wp
    if dg.any():
        assert np.all(r["status"][dg] >= 0)
        conv = dg & (r["status"] == 0)
        assert np.all(r["gnorm"][conv] <= 1e-5)
    check_d_class(b, opts, exp, r, name)


def test_exit_fixtures_cover_every_lbfgsb_branch():
    """The exit_* fixtures (generate_exit_fixtures.py) are there for the stops and branches ordinary entities never take. On
    their strict entities the oracle's trajectory IS the reference's (identical nit / nfev / status asserted above), so
    the oracle's branch counters say what the reference went through: FACTR and ABNORMAL stops, the curvature-skip rule,
    the g'd >= 0 restart, the maxls abort with and without history, searches of more than ten evaluations."""
    names = [n for n in fixture_names() if n.startswith("exit_")]
    assert names
    hist = np.zeros(5, np.int64)
    total = dict(skipped_pairs=0, gd_restarts=0, maxls_aborts=0, memory_wraps=0)
    longest = 0
    abort_with_history = 0
    for name in names:
        b, opts, exp, _ = load_fixture(name)
        st = np.flatnonzero(exp["strict"].astype(bool))
        sub = b.select(st)
        pk = oracle.pack(sub.ent_row_ptr, sub.row_nnz_ptr, sub.col_global)
        oracle.branch_counts()
        r = oracle.solve(pk, sub.val, sub.y, sub.offset, sub.weight, oracle.make_opts(**opts_kwargs(opts)))
        bc = oracle.branch_counts()
        assert np.array_equal(r["nit"], exp["nit"][st]) and np.array_equal(r["nfev"], exp["nfev"][st])
        assert np.array_equal(r["status"], exp["status"][st])
        hist += np.bincount(exp["status"][st], minlength=5)[:5]
        for k in total:
            total[k] += bc[k]
        longest = max(longest, bc["max_evals_in_one_search"])
        abort_with_history += int(((exp["status"][st] != 4) & (exp["nfev"][st] - exp["nit"][st] > 20)).sum())
    assert hist[0] > 0 and hist[1] > 0 and hist[2] > 0 and hist[4] > 0, hist        # PGTOL, FACTR, MAXITER, ABNORMAL
    assert total["skipped_pairs"] > 0 and total["gd_restarts"] > 0 and total["maxls_aborts"] > 0 and total["memory_wraps"] > 0, total
    assert longest == 20 and abort_with_history > 0


def _mask(coef_ptr, ent_mask):
    m = np.zeros(coef_ptr[-1], bool)
    for e in np.flatnonzero(ent_mask):
        m[coef_ptr[e]:coef_ptr[e + 1]] = True
    return m


def test_oracle_score_matches_reference_inference():
    b, opts, exp, _ = load_fixture("warm_stage2")
    z = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "score_stage2.npz"))
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    has_model = np.zeros(b.E, np.uint8)
    has_model[:int(z["n_with_model"])] = 1
    # models are the thresholded coefficients the training consumer returned
    logit, per = oracle.score(pk, b.val, b.offset, exp["theta_thr"], True, has_model)
    assert np.array_equal(z["exp_uid"], b.uid)
    np.testing.assert_allclose(logit, z["exp_score"].astype(np.float32), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(per, z["exp_per_coord"].astype(np.float32), rtol=1e-5, atol=1e-6)
    # entities without a model: logit == offset exactly, per-coordinate score == 0
    r0 = b.ent_row_ptr[int(z["n_with_model"])]
    assert np.array_equal(logit[r0:], b.offset[r0:])
    assert np.all(per[r0:] == 0)


# hand-derived known answers for Math.abs(s.hashCode) % n
JAVA_HASH_KAT = [("0", 48), ("100034", 1448635136), ("abc102", -1424436655),
                 ("polygenelubricants", -2147483648), ("Aa", 2112), ("BB", 2112), ("\U0001F600", 1772899), ("", 0)]
JAVA_PART_KAT = [("0", 10, 8), ("1", 10, 9), ("12", 10, 9), ("100", 10, 5), ("943", 10, 0), ("1682", 10, 9),
                 ("polygenelubricants", 10, -8)]


@pytest.mark.parametrize("s,h", JAVA_HASH_KAT)
def test_java_string_hash_kat(s, h):
    assert oracle.java_string_hash(s) == h


@pytest.mark.parametrize("s,n,pid", JAVA_PART_KAT)
def test_java_partition_id_kat(s, n, pid):
    assert oracle.java_partition_id(s, n) == pid
