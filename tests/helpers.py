"""Shared helpers for the parity tests: golden-fixture loading and parity-class bookkeeping."""
import glob
import json
import os

import numpy as np

from gdmix_amd.batch import RawBatch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# theta parity bar of BASELINE.json's north_star: 1e-5 relative (max-norm, per entity) on well-posed
# entities. With fp64 solver state the restatement is expected to sit many orders below it.
REL_TOL_NORTH_STAR = 1e-5
REL_TOL_ORACLE = 1e-8      # CPU restatement vs reference scipy fixtures (well-posed class)


def fixture_names(solve_only=True):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    names = [n for n in names if not n.startswith("fe_")]   # fixed-effect fixtures: tests/test_fixed_effect.py
    if solve_only:
        names = [n for n in names if not n.startswith("score_")]
    return names


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    opts = json.loads(str(z["opts"])) if "opts" in z.files else {}
    batch = None
    if "ent_row_ptr" in z.files:
        batch = RawBatch(ent_row_ptr=z["ent_row_ptr"], row_nnz_ptr=z["row_nnz_ptr"], col_global=z["col_global"],
                         val=z["val"], y=z["y"], offset=z["offset"],
                         weight=z["weight"] if "weight" in z.files else None, uid=z["uid"],
                         entity_ids=[str(s) for s in z["exp_entity_ids"]])
    exp = {k[4:]: z[k] for k in z.files if k.startswith("exp_")}
    extra = {k: z[k] for k in z.files if k.startswith("prior_") or k in ("n_with_model",)}
    return batch, opts, exp, extra


VAR_MODE = {None: 0, "simple": 1, "full": 2}


def opts_kwargs(opts):
    """Fixture opts json -> kwargs shared by oracle.make_opts and gdmix_amd SolverOptions."""
    return dict(l2=opts["l2"], regularize_bias=opts["regularize_bias"], has_intercept=opts["has_intercept"],
                m=opts["m"], max_iter=opts["max_iter"], ftol=opts["tol"],
                variance_mode=VAR_MODE[opts.get("variance_mode")])


def well_posed_mask(batch, opts):
    """Parity class W of SURVEY.md §8(d): both labels present, or the intercept is regularised with
    l2 > 0 (or there is no intercept and l2 > 0). Class D entities have no finite optimum and the
    reference's own answer there is chaotic; only invariants are checked for them."""
    E = batch.E
    n1 = np.add.reduceat(batch.y, batch.ent_row_ptr[:-1]) if batch.N else np.zeros(E)
    n = batch.ent_n()
    mixed = (n1 > 0) & (n1 < n)
    if opts["l2"] > 0 and (opts["regularize_bias"] or not opts["has_intercept"]):
        return np.ones(E, bool)
    return mixed


def parity_mask(batch, opts, exp):
    """Entities compared iteration for iteration. Fixtures of tests/golden/generate_exit_fixtures.py carry `strict`: the
    reference reproduced its own answer there under rounding-sized noise (start +-1e-14, weights +-1 ulp); the others of
    those sets are long runs on extreme inputs whose reference answer depends on scipy's build. Older fixtures: class W."""
    if "strict" in exp:
        return exp["strict"].astype(bool)
    return well_posed_mask(batch, opts)


def entity_logits(batch, e, uniq_e, theta_e, ic):
    """x_i . theta + offset_i for the samples of entity e; theta_e in local index space (intercept first)."""
    r0, r1 = int(batch.ent_row_ptr[e]), int(batch.ent_row_ptr[e + 1])
    z = batch.offset[r0:r1].astype(np.float64) + (theta_e[0] if ic else 0.0)
    for i in range(r0, r1):
        k0, k1 = int(batch.row_nnz_ptr[i]), int(batch.row_nnz_ptr[i + 1])
        loc = np.searchsorted(uniq_e, batch.col_global[k0:k1])
        z[i - r0] += float(np.dot(batch.val[k0:k1].astype(np.float64), theta_e[ic + loc]))
    return z


def d_class_contract(batch, opts, feat_ptr, uniq, theta, theta_thr, gnorm, status, entities):
    """SURVEY.md §8(d) class D (all labels equal, intercept unregularised: no finite optimum): per entity, which of the four
    invariants hold — (0) returned max|g| <= 1e-5, (1) sign(theta_0) matches the label, (2) every non-intercept
    |theta_j| <= 1e-4, i.e. thresholded to 0, (3) sigmoid(logit) within 1e-4 of the label on the entity's own samples.
    Returns a bool array [len(entities), 4]."""
    ic = 1 if opts["has_intercept"] else 0
    out = np.zeros((len(entities), 4), bool)
    for r, e in enumerate(entities):
        f0, f1 = int(feat_ptr[e]), int(feat_ptr[e + 1])
        c0 = f0 + e * ic
        th = theta[c0:c0 + (f1 - f0) + ic]
        label = float(batch.y[int(batch.ent_row_ptr[e])])
        out[r, 0] = status[e] == 0 and gnorm[e] <= 1e-5
        out[r, 1] = bool(ic) and ((th[0] > 0) == (label > 0.5)) and th[0] != 0
        out[r, 2] = bool(np.all(np.abs(th[ic:]) <= 1e-4)) and bool(np.all(theta_thr[c0 + ic:c0 + (f1 - f0) + ic] == 0))
        z = entity_logits(batch, e, uniq[f0:f1], th, ic)
        with np.errstate(over="ignore"):
            out[r, 3] = bool(np.all(np.abs(1.0 / (1.0 + np.exp(-z)) - label) <= 1e-4))
    return out


def check_d_class(batch, opts, exp, res, name=""):
    """Device / oracle result `res` (dict of host arrays) on the class-D entities of a fixture: every invariant the
    reference's own answer satisfies must hold for ours too, and the thresholded zero pattern must be the fixture's
    whenever the non-intercept coefficients are all thresholded away. Returns (#D entities, #with the full contract)."""
    dg = np.flatnonzero(~well_posed_mask(batch, opts))
    if dg.size == 0 or not opts["has_intercept"]:
        return 0, 0
    fp, uq = exp["ent_feat_ptr"], exp["unique_global"]
    want = d_class_contract(batch, opts, fp, uq, exp["theta"], exp["theta_thr"], exp["gnorm"], exp["status"], dg)
    got = d_class_contract(batch, opts, fp, uq, res["theta"], res["theta_thr"], res["gnorm"], res["status"], dg)
    bad = want & ~got
    assert not bad.any(), f"{name}: class-D invariants lost on entities {dg[bad.any(axis=1)][:8].tolist()} (columns {np.flatnonzero(bad.any(axis=0)).tolist()})"
    assert np.all(res["status"][dg] >= 0)
    return int(dg.size), int(want.all(axis=1).sum())


def per_entity_rel_err(a, b, coef_ptr):
    """max-norm relative error per entity: max|a-b| / max(max|b|, tiny)."""
    E = coef_ptr.size - 1
    out = np.zeros(E)
    for e in range(E):
        s = slice(coef_ptr[e], coef_ptr[e + 1])
        den = max(np.max(np.abs(b[s])) if coef_ptr[e + 1] > coef_ptr[e] else 0.0, 1e-300)
        out[e] = (np.max(np.abs(a[s] - b[s])) if coef_ptr[e + 1] > coef_ptr[e] else 0.0) / den
    return out


# ---- CPU test double of REDeviceSolver built on the oracle (tests only) ---------------------------------
class _Arr:
    """numpy array with the .cpu().numpy() surface of a torch tensor."""

    def __init__(self, a):
        self.a = a

    def cpu(self):
        return self

    def to(self, dtype):      # (model.py widens the device's int32 feature map before the copy; the oracle's is int64 already)
        return self

    def numpy(self):
        return self.a


class OraclePacked:
    def __init__(self, pk, batch, has_intercept):
        self.pk, self.batch, self.has_intercept = pk, batch, has_intercept
        self.E, self.N, self.Z, self.D = pk["E"], pk["N"], pk["Z"], pk["D"]
        self.P = self.D + (self.E if has_intercept else 0)

    def ent_feat_ptr(self):
        return _Arr(self.pk["ent_feat_ptr"])

    def unique_global(self):
        return _Arr(self.pk["unique_global"])

    def coef_ptr_host(self):
        return self.pk["ent_feat_ptr"] + (np.arange(self.E + 1) if self.has_intercept else 0)


class _Res:
    def __init__(self, d):
        self.d = d

    def to_host(self, keys=None):
        return self.d if keys is None else {k: v for k, v in self.d.items() if k in keys}


class OracleSolverDouble:
    """Stands in for gdmix_amd.solver.REDeviceSolver in CPU-only tests of the host logic (model, driver,
    I/O). It routes pack/solve/score through oracle/ — allowed for tests, never for the product path."""

    def pack(self, batch, has_intercept=True):
        from oracle import oracle
        return OraclePacked(oracle.pack(batch.ent_row_ptr, batch.row_nnz_ptr, batch.col_global), batch, has_intercept)

    def solve(self, packed, opts, theta0=None, out=None):
        from oracle import oracle
        b = packed.batch
        o = oracle.make_opts(l2=opts.l2, regularize_bias=opts.regularize_bias, has_intercept=opts.has_intercept, m=opts.m,
                             max_iter=opts.max_iter, ftol=opts.ftol, variance_mode=int(opts.variance_mode),
                             threshold=opts.threshold)
        return _Res(oracle.solve(packed.pk, b.val, b.y, b.offset, b.weight, o, theta0=theta0))

    def score(self, packed, theta, has_model=None):
        from oracle import oracle
        b = packed.batch
        lo, pc = oracle.score(packed.pk, b.val, b.offset, theta, packed.has_intercept, has_model)
        return _Arr(lo), _Arr(pc)


class OracleFeDouble:
    """Stands in for gdmix_amd.fixed_effect.FixedEffectDeviceSolver in CPU-only tests of the fixed-effect model class:
    same fit_stepping() contract, solved by the oracle; `.solver` is the OracleSolverDouble (pack / score)."""

    def __init__(self):
        self.solver = OracleSolverDouble()

    def fit_stepping(self, row_nnz_ptr, col_global, val, y, num_features, offset=None, weight=None, has_intercept=True, l2=1.0,
                     regularize_bias=True, model_type="logistic_regression", theta0=None, max_iter=100, m=10, tolerance=1e-12,
                     group=None, dummy=None, variance_mode=None, threshold=0.0):
        from gdmix_amd import fixed_effect as fe
        from oracle import oracle
        linear = model_type == fe.LINEAR_REGRESSION
        batch, dummy = fe.shard_as_batch(row_nnz_ptr, col_global, val, y, offset, weight, has_intercept, binary_labels=not linear,
                                         dummy=dummy)
        D = 1 if dummy else int(num_features)
        pk = oracle.pack(batch.ent_row_ptr, batch.row_nnz_ptr, batch.col_global)
        uniq = pk["unique_global"]
        ic = 1 if has_intercept else 0
        t0 = None
        if theta0 is not None:
            full = np.zeros(D + ic)
            th = np.asarray(theta0, np.float64)
            if dummy:
                full[D:] = th[-ic:] if ic else []
            else:
                full[:] = th
            t0 = fe.to_local(full, uniq, D, has_intercept, False)
        o = oracle.make_opts(l2=l2, regularize_bias=bool(regularize_bias) and has_intercept, has_intercept=has_intercept, m=m,
                             max_iter=max_iter, ftol=tolerance, threshold=0.0, sum_loss=True, linear=linear)
        res = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o, theta0=t0)
        theta = fe.to_global(res["theta"], uniq, D, has_intercept, False)
        variances = None
        if variance_mode is not None:
            class _NoDevice:    # FULL runs on the host; SIMPLE stated here in numpy (the device pass is tested on the GPU)
                pass
            th = np.where(np.abs(theta) <= threshold, 0.0, theta)
            rb = bool(regularize_bias) and has_intercept
            if str(variance_mode).upper() == "FULL":
                variances = fe._variances(_NoDevice(), None, batch, th, D, has_intercept, float(l2), rb, "FULL", None, None)
            else:
                k = np.diff(batch.row_nnz_ptr)
                rows = np.repeat(np.arange(batch.N), k)
                z = np.bincount(rows, weights=batch.val.astype(np.float64) * th[batch.col_global], minlength=batch.N) + batch.offset
                if has_intercept:
                    z = z + th[D]
                rho = 1.0 / (1.0 + np.exp(-z))
                d = rho * (1 - rho) * (1.0 if batch.weight is None else batch.weight)
                H = np.bincount(batch.col_global, weights=batch.val.astype(np.float64) ** 2 * d[rows], minlength=D)
                if has_intercept:
                    H = np.concatenate([H, [d.sum()]])
                H = H + l2
                if has_intercept and not rb:
                    H[-1] -= l2
                variances = 1.0 / (H + 1e-12)
        if dummy:
            theta = theta[D:]
            if variances is not None:
                variances = variances[D:]
        extra = {} if variances is None else {"variances": variances}
        return theta, dict(**extra, fval=res["fval"][0], gnorm=res["gnorm"][0], nit=int(res["nit"][0]), nfev=int(res["nfev"][0]),
                           status=int(res["status"][0]))


# ---- full-size property tests: a class-stratified sample against the oracle ------------------------------
def oracle_solve_parallel(b, kw, theta0=None):
    """oracle.pack + oracle.solve of a host RawBatch with the entities dealt to the host cores in contiguous runs of about
    equal non-zero count (ctypes releases the GIL). Returns (pk, result dict)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    o = oracle.make_opts(**kw)
    cores = min(os.cpu_count() or 1, max(1, b.E))
    work = np.concatenate([[0], np.cumsum(b.ent_nnz() + 64)])
    bounds = np.searchsorted(work, np.linspace(0, work[-1], cores + 1)).astype(int)
    bounds[0], bounds[-1] = 0, b.E
    ic = 1 if kw.get("has_intercept", True) else 0
    cp = pk["ent_feat_ptr"] + np.arange(b.E + 1) * ic

    def run(i):
        e0, e1 = int(bounds[i]), int(bounds[i + 1])
        if e0 >= e1:
            return None
        r = oracle.solve(pk, b.val, b.y, b.offset, b.weight, o, theta0=theta0, e_begin=e0, e_end=e1)
        return e0, e1, {k: (v[cp[e0]:cp[e1]].copy() if k in ("theta", "theta_thr", "variance") else v[e0:e1].copy())
                        for k, v in r.items() if v is not None}
    with ThreadPoolExecutor(cores) as ex:
        parts = [p for p in ex.map(run, range(cores)) if p is not None]
    out = {}
    for k in parts[0][2]:
        out[k] = np.concatenate([p[2][k] for p in parts])
    return pk, out


def stratified_sample(cls, nnz, rng, total=2000, nnz_budget_per_class=6_000_000):
    """Entity indices for an oracle comparison, stratified by kernel class: every class with at least one entity gives
    min(count, quota) entities drawn at random, stopping early once the class's sample holds `nnz_budget_per_class` non-zeros
    (the first draw is always taken, so the classes of giant entities are sampled too). Returns (sorted indices, {class: taken})."""
    present = np.flatnonzero(np.bincount(cls, minlength=1))
    quota = max(1, total // max(1, present.size))
    take, taken = [], {}
    for c in present:
        members = np.flatnonzero(cls == c)
        members = members[rng.permutation(members.size)][:quota]
        csum = np.cumsum(nnz[members])
        keep = max(1, int(np.searchsorted(csum, nnz_budget_per_class, side="right")))
        take.append(members[:keep])
        taken[int(c)] = int(min(keep, members.size))
    take = np.concatenate(take)
    if take.size < total:    # fill up from the bulk (entities of ordinary size, any class)
        rest = np.setdiff1d(np.flatnonzero(nnz <= 20_000), take)
        take = np.concatenate([take, rest[rng.permutation(rest.size)][:total - take.size]])
    return np.sort(take), taken
