"""CPU restatement of one pass of the coordinate chain (gdmix_amd/chain.py) — test infrastructure, never imported by the product:
global fixed effect -> per-user random effect -> per-movie random effect on the SAME flat data set, every solve by the fp64 oracle
(oracle/re_oracle.c, pinned to the reference by tests/golden), every score truncated to float32 exactly where the Avro score files
do (`predictionScore` is an Avro float: util/io_utils.py:367-375; OffsetUpdater casts it to FLOAT: OffsetUpdater.scala:115-116), the
next stage's offset = that float joined on uid, models thresholded at 1e-4 before scoring (job_consumers.py:62,
fixed_effect_lr_lbfgs_model.py:648-649).
"""
import numpy as np

from gdmix_amd import chain
from gdmix_amd import fixed_effect as fe
from oracle import oracle

THRESHOLD = 1e-4


def _threshold(theta):
    return np.where(np.abs(theta) <= THRESHOLD, 0.0, theta)


def _dense_scores(ptr, cols, vals, coef, icpt, offset):
    """float64 row sums intercept + sum(val * coef[col]) per sample (coef: [n_samples-aligned rows x dim] looked up by the caller)."""
    rows = np.repeat(np.arange(ptr.size - 1), np.diff(ptr))
    z = icpt + np.bincount(rows, vals.astype(np.float64) * coef, ptr.size - 1)
    per = z.astype(np.float32)
    return (z + offset.astype(np.float64)).astype(np.float32), per


def global_stage(data):
    """-> theta [D_GLOBAL + 1] (intercept last, thresholded), {uid: score} for training and validation samples."""
    tr = np.flatnonzero(data["train"])
    ptr, cols, vals, dim = chain.bag_rows(data, "global", tr)
    y = data["response"][tr].astype(np.float32)
    batch, dummy = fe.shard_as_batch(ptr, cols, vals, y, None, None, True)
    pk = oracle.pack(batch.ent_row_ptr, batch.row_nnz_ptr, batch.col_global)
    o = oracle.make_opts(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, threshold=0.0, sum_loss=True)
    res = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o)
    theta = _threshold(fe.to_global(res["theta"], pk["unique_global"], dim, True, dummy))
    out = {"theta": theta, "info": {k: int(res[k][0]) for k in ("nit", "nfev", "status")}}
    for name, rows in (("train", tr), ("validation", np.flatnonzero(~data["train"]))):
        p, c, v, _ = chain.bag_rows(data, "global", rows)
        off = np.zeros(rows.size, np.float32)
        score, per = _dense_scores(p, c, v, theta[c], theta[dim], off)
        # the fixed-effect model writes score = float32(float64(per-coordinate float32) + offset) (fe_model._score_and_write)
        score = (per.astype(np.float64) + off.astype(np.float64)).astype(np.float32)
        out[name] = {"uid": data["uid"][rows], "score": score, "per_coord": per}
    return out


def random_effect_stage(data, stage, prev, upper_bound=None):
    """prev: the previous stage's {"train": {uid, score}, "validation": {...}}. -> per-entity thresholded coefficients in the
    global index space (dict entity -> (intercept, dense [dim])), W-class flags, and the stage's scores."""
    ent_all = data["user"] if stage == "per_user" else data["movie"]
    dim = data["bags"][stage][3]
    tr = np.flatnonzero(data["train"])
    order = np.argsort(prev["train"]["uid"], kind="stable")
    pos = np.searchsorted(prev["train"]["uid"][order], data["uid"][tr])
    off_tr = prev["train"]["score"][order][pos].astype(np.float32)
    # active data: with an upper bound an entity of `count` training samples is cut into count / upper_bound + 1 groups by
    # uid mod groups and trained on group 0 only (DataPartitioner.scala:358-376); everything is scored
    active = np.ones(tr.size, bool)
    if upper_bound is not None:
        _, inv, cnt = np.unique(ent_all[tr], return_inverse=True, return_counts=True)
        groups = (cnt[inv] / float(upper_bound) + 1.0).astype(np.int64)
        active = np.mod(data["uid"][tr], groups) == 0
    # entity-major, samples in input order
    grp = np.argsort(ent_all[tr][active], kind="stable")
    rows = tr[active][grp]
    off_tr = off_tr[active]
    ents, first, counts = np.unique(ent_all[rows], return_index=True, return_counts=True)
    ent_row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ptr, cols, vals, _ = chain.bag_rows(data, stage, rows)
    y = data["response"][rows].astype(np.float32)
    off = off_tr[grp]
    pk = oracle.pack(ent_row_ptr, ptr, cols)
    o = oracle.make_opts(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, threshold=THRESHOLD)
    res = oracle.solve(pk, vals, y, off, None, o)
    E = ents.size
    icpt = np.zeros(E)
    coef = np.zeros((E, dim))
    fp = pk["ent_feat_ptr"]
    for e in range(E):     # coefficients of entity e: intercept, then its features in ascending global index
        base = fp[e] + e
        icpt[e] = res["theta_thr"][base]
        coef[e, pk["unique_global"][fp[e]:fp[e + 1]]] = res["theta_thr"][base + 1:base + 1 + fp[e + 1] - fp[e]]
    ones = np.add.reduceat(y.astype(np.float64), ent_row_ptr[:-1])
    out = {"entities": ents, "intercept": icpt, "coef": coef, "well_posed": (ones > 0) & (ones < counts), "active_samples": int(active.sum()),
           "status": res["status"], "nit": res["nit"], "raw_theta": res["theta"], "feat_ptr": fp, "unique_global": pk["unique_global"]}
    for name, mask, prev_s in (("train", data["train"], prev["train"]), ("validation", ~data["train"], prev["validation"])):
        r = np.flatnonzero(mask)
        order = np.argsort(prev_s["uid"], kind="stable")
        pos = np.searchsorted(prev_s["uid"][order], data["uid"][r])
        offs = prev_s["score"][order][pos].astype(np.float32)
        p, c, v, _ = chain.bag_rows(data, stage, r)
        e_idx = np.searchsorted(ents, ent_all[r])
        has = (e_idx < E) & (ents[np.minimum(e_idx, E - 1)] == ent_all[r])
        e_idx = np.where(has, e_idx, 0)
        e_rows = np.repeat(e_idx, np.diff(p))
        h_rows = np.repeat(has, np.diff(p))
        score, per = _dense_scores(p, c, v, np.where(h_rows, coef[e_rows, c], 0.0), np.where(has, icpt[e_idx], 0.0), offs)
        # samples of entities without a finite optimum (all labels equal, intercept unregularised): their scores are comparable only
        # through the sigmoid (SURVEY 8(d) class D)
        out[name] = {"uid": data["uid"][r], "score": score, "per_coord": per, "offset": offs, "strict": ~has | out["well_posed"][e_idx]}
    return out


def run(data, upper_bounds=None):
    ub = upper_bounds or {}
    g = global_stage(data)
    u = random_effect_stage(data, "per_user", g, ub.get("per_user"))
    m = random_effect_stage(data, "per_movie", u, ub.get("per_movie"))
    return {"global": g, "per_user": u, "per_movie": m}
