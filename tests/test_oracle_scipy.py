"""The oracle against the reference's optimiser itself, run here: scipy.optimize.fmin_l_bfgs_b (what binary_logistic_regression.py:223-231 and
fixed_effect_lr_lbfgs_model.py:635-643 call) on the objective restated in numpy, seeded shards of tools/fuzz_fe.py's generator. The golden
fixtures pin the oracle to outputs of the reference generated once; this pins it to L-BFGS-B's behaviour on fresh problems — status,
iterations, evaluations and the coefficients — including the case that showed the device's compact-form defect in round 6. CPU only.
(scipy here is 1.15, whose L-BFGS-B is a C translation of the 3.0 Fortran the reference's pinned 1.5.4 wraps: same algorithm.)"""
import numpy as np
import pytest

from gdmix_amd import fixed_effect as fe
from oracle import oracle
import fuzz_fe_case

sp = pytest.importorskip("scipy.sparse")
opt = pytest.importorskip("scipy.optimize")


def objective(c, pk, sum_loss):
    """value and gradient in the oracle's local space (intercept first, then the features present), as the reference forms them:
    logistic max(z,0) - z y + log(1 + exp(-|z|)) or squared_difference, weighted; l2 / 2 on the regularised coefficients; the random
    effect divides everything by the number of samples (sum_loss = False)."""
    uq = np.asarray(pk["unique_global"])
    nf, i0 = uq.size, (1 if c.ic else 0)
    loc = np.full(c.D, -1, np.int64)
    loc[uq] = np.arange(nf)
    X = sp.csr_matrix((c.vals.astype(np.float64), loc[c.cols], c.rp), shape=(c.n, nf))
    offs = np.zeros(c.n) if c.off is None else c.off.astype(np.float64)
    w = np.ones(c.n) if c.wt is None else c.wt.astype(np.float64)
    y = c.y.astype(np.float64)
    reg = np.ones(nf + i0)
    if c.ic and not c.regb:
        reg[0] = 0.0
    scale = 1.0 if sum_loss else 1.0 / c.n

    def fg(th):
        z = X @ th[i0:] + offs + (th[0] if c.ic else 0.0)
        if c.linear:
            r = z - y
            f, gr = np.sum(w * r * r), 2.0 * w * r
        else:
            f, gr = np.sum(w * (np.maximum(z, 0.0) - z * y + np.log1p(np.exp(-np.abs(z))))), w * (1.0 / (1.0 + np.exp(-z)) - y)
        g = np.empty(nf + i0)
        g[i0:] = X.T @ gr
        if c.ic:
            g[0] = gr.sum()
        return scale * (f + 0.5 * c.l2 * np.sum(reg * th * th)), scale * (g + c.l2 * reg * th)
    return fg, nf + i0


def small_seeds(first, count, max_z=400_000):
    out, s = [], first
    while len(out) < count:
        c = fuzz_fe_case.draw(s)
        if c.Z <= max_z and c.Z > 0 and c.th0 is None:
            out.append(s)
        s += 1
    return out


SEEDS = small_seeds(7100000, 10)


@pytest.mark.parametrize("sum_loss", [True, False])
@pytest.mark.parametrize("seed", SEEDS + [6700230])
def test_oracle_takes_scipys_path(seed, sum_loss):
    c = fuzz_fe_case.draw(seed)
    if seed == 6700230 and not sum_loss:
        pytest.skip("the round-6 case is a fixed-effect fit")
    batch, dummy = fe.shard_as_batch(c.rp, c.cols, c.vals, c.y, c.off, c.wt, c.ic, binary_labels=not c.linear)
    if dummy:
        pytest.skip("a shard without a non-zero trains on a stand-in column")
    pk = oracle.pack(batch.ent_row_ptr, batch.row_nnz_ptr, batch.col_global)
    fg, P = objective(c, pk, sum_loss)
    tol = 1e-12 if sum_loss else 1e-7      # the fixed effect's lbfgs_tolerance; the random effect's default
    x, f, info = opt.fmin_l_bfgs_b(fg, np.zeros(P), m=c.m, factr=tol / np.finfo(float).eps, pgtol=1e-5, maxiter=c.max_iter)
    o = oracle.make_opts(l2=c.l2, regularize_bias=c.regb and c.ic, has_intercept=c.ic, m=c.m, max_iter=c.max_iter, ftol=tol, threshold=0.0,
                         sum_loss=sum_loss, linear=c.linear)
    res = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o)
    theta = np.asarray(res["theta"]).ravel()
    assert theta.shape == x.shape
    # scipy: warnflag 0 = converged (either test), 1 = iteration / evaluation limit, 2 = abnormal; the oracle: 0 pgtol, 1 factr, 2 max_iter
    want = {0: (0, 1), 1: (2, 3), 2: (4,)}[info["warnflag"]]
    assert int(res["status"][0]) in want, (info["task"], int(res["status"][0]))
    err = float(np.max(np.abs(theta - x)) / max(np.max(np.abs(x)), 1e-300))
    counts = (int(res["nit"][0]), int(res["nfev"][0]))
    if int(res["status"][0]) == 1 and counts != (info["nit"], info["funcalls"]):
        # a factr stop is decided at the objective's rounding level — (f_old - f) <= tol |f| — and the numpy objective adds in another order than the
        # oracle: after dozens of iterations the test may fire a few iterations apart (tests/test_fixed_effect.py: linear_wide, 77 iterations).
        # Then the counts are close and both stand at the same value; how far apart the coefficients are is the problem's conditioning
        # (seed 7100018: five features, 1e-4 — the oracle moves as much under a 1e-13 change of its start: tools/fuzz_fe.py's `sensitive`).
        assert info["nit"] >= 30 and abs(counts[0] - info["nit"]) <= max(3, info["nit"] // 10), (info["task"], counts, info["nit"])
        assert abs(float(res["fval"][0]) - f) <= 1e-9 * max(1.0, abs(f)) and err <= 1e-3, (float(res["fval"][0]), f, err)
        return
    assert counts == (info["nit"], info["funcalls"]), (info["task"], counts, info["nit"], info["funcalls"])
    assert abs(float(res["fval"][0]) - f) <= (1e-9 if int(res["status"][0]) == 1 else 1e-11) * max(1.0, abs(f))
    assert err <= (1e-5 if int(res["status"][0]) == 1 else 1e-8), err      # a factr stop sits at rounding level (tests/test_fixed_effect.py: REL_TOL_FACTR)
