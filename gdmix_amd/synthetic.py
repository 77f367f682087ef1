"""Seeded synthetic entity-grouped data of the shapes BASELINE.json / SURVEY.md §8(d) name.

Every entity gets n_e samples; every sample gets k distinct global columns (one per stratum of the
global feature space, so they are distinct and ascending by construction), fp32 values ~ N(0,1), an
fp32 offset ~ N(0,1) and a label y ~ Bernoulli(sigmoid(x . w* + b_e + offset)) with a hidden global
w* ~ 0.5 N(0,1) and a hidden per-entity bias b_e ~ 0.5 N(0,1). Sample weights are 1 unless asked.

There is no network in the build or GPU containers, so MovieLens itself cannot be downloaded; the
"ml_*" shapes reproduce the per-user / per-movie feature bags of the reference's preprocessing script
(scripts/download_process_movieLens_data.py:306-346,384-387): per_user = 1-6 genre flags (value 1.0)
out of 19 + release_date/2000 (D = 20); per_movie = age/100 + gender one-hot + occupation one-hot
(k = 3, D = 24).
"""
import numpy as np

from .batch import RawBatch

C2_SEED = 20240601
C5_SEED = 20240605


def _entity_sizes(rng, E, shape, mean_n):
    if shape == "poisson":
        return np.maximum(1, rng.poisson(mean_n, size=E)).astype(np.int64)
    if shape == "const":
        return np.full(E, int(mean_n), np.int64)
    if shape == "geometric":
        return rng.geometric(1.0 / mean_n, size=E).astype(np.int64)
    if shape == "zipf":
        # P(n >= x) ~ x^-1.2 truncated, rescaled to the requested mean
        u = rng.random(E)
        raw = np.minimum((1.0 - u) ** (-1.0 / 1.2), 2.0 ** 17)
        n = np.maximum(1, np.floor(raw * mean_n / raw.mean())).astype(np.int64)
        return n
    raise ValueError(f"unknown size distribution {shape!r}")


def make_batch(E, mean_n=16, k=4, D=1024, seed=C2_SEED, size_dist="poisson", l_offset=1.0,
               value_scale=1.0, random_weights=False, entity_id_base=0, with_uid=True):
    """Generic sparse-bag entities (C2: E=1e6, mean_n=16, k=4, D=1024; C5 mean shape: 32, 8, 65536)."""
    rng = np.random.default_rng(seed)
    n = _entity_sizes(rng, E, size_dist, mean_n)
    N = int(n.sum())
    if D % k:
        raise ValueError("D must be a multiple of k")
    stride = D // k
    cols = (rng.integers(0, stride, size=(N, k), dtype=np.int64) + np.arange(k, dtype=np.int64) * stride)
    vals = (rng.standard_normal(size=(N, k)) * value_scale).astype(np.float32)
    offset = (rng.standard_normal(N) * l_offset).astype(np.float32)
    w_star = 0.5 * rng.standard_normal(D)
    b_e = 0.5 * rng.standard_normal(E)
    logit = (vals.astype(np.float64) * w_star[cols]).sum(axis=1) + np.repeat(b_e, n) + offset
    y = (rng.random(N) < 1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
    weight = (0.25 + 2.0 * rng.random(N)).astype(np.float32) if random_weights else None
    return RawBatch(
        ent_row_ptr=np.concatenate([[0], np.cumsum(n)]),
        row_nnz_ptr=np.arange(N + 1, dtype=np.int64) * k,
        col_global=cols.reshape(-1), val=vals.reshape(-1), y=y, offset=offset, weight=weight,
        uid=np.arange(N, dtype=np.int64) if with_uid else None,
        entity_ids=[str(i) for i in range(entity_id_base, entity_id_base + E)])


def make_survey_batch(E, mean_n=16, k=4, D=1024, seed=C2_SEED, size_dist="poisson", entity_id_base=0, with_uid=False):
    """The generator SURVEY.md §8(d) states for the measured configurations, to the letter: n_e = max(1, Poisson(mean_n)); per
    sample k DISTINCT columns drawn uniformly from [0, D) (in draw order, not sorted); values ~ N(0,1) fp32; offset ~ N(0,1)
    fp32; hidden w* ~ 0.5 N(0,1); y ~ Bernoulli(sigmoid(x . w* + offset)) — no per-entity bias; weight = 1.
    (make_batch above — one column per stratum of the feature space plus a hidden entity bias — is what the committed
    parity fixtures were generated from and stays as it is.)"""
    rng = np.random.default_rng(seed)
    n = _entity_sizes(rng, E, size_dist, mean_n)
    N = int(n.sum())
    cols = rng.integers(0, D, size=(N, k), dtype=np.int64)
    while k > 1:   # re-draw the samples that drew a column twice (k = 4, D = 1024: 0.6 % of them)
        dup = np.zeros(N, bool)
        for a in range(k):
            for b in range(a + 1, k):
                dup |= cols[:, a] == cols[:, b]
        bad = np.flatnonzero(dup)
        if bad.size == 0:
            break
        cols[bad] = rng.integers(0, D, size=(bad.size, k), dtype=np.int64)
    vals = rng.standard_normal(size=(N, k)).astype(np.float32)
    offset = rng.standard_normal(N).astype(np.float32)
    w_star = 0.5 * rng.standard_normal(D)
    logit = (vals.astype(np.float64) * w_star[cols]).sum(axis=1) + offset
    y = (rng.random(N) < 1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
    return RawBatch(
        ent_row_ptr=np.concatenate([[0], np.cumsum(n)]), row_nnz_ptr=np.arange(N + 1, dtype=np.int64) * k,
        col_global=cols.reshape(-1), val=vals.reshape(-1), y=y, offset=offset, weight=None,
        uid=np.arange(N, dtype=np.int64) if with_uid else None,
        entity_ids=[str(i) for i in range(entity_id_base, entity_id_base + E)])


def make_ragged_batch(E, seed=7, D=200, max_n=40, max_k=9, empty_row_prob=0.15, dup_prob=0.1,
                      random_weights=True):
    """Adversarially ragged entities: empty samples, duplicate (row, col) pairs, n from 1, k from 0."""
    rng = np.random.default_rng(seed)
    n = rng.integers(1, max_n + 1, size=E).astype(np.int64)
    N = int(n.sum())
    k = rng.integers(1, max_k + 1, size=N).astype(np.int64)
    k[rng.random(N) < empty_row_prob] = 0
    # the reference requires the last sample of an entity to carry a feature (job_consumers.py:229-232)
    last = np.cumsum(n) - 1
    k[last] = np.maximum(k[last], 1)
    Z = int(k.sum())
    cols = rng.integers(0, D, size=Z, dtype=np.int64)
    dup = np.flatnonzero(rng.random(Z) < dup_prob)
    row_of = np.repeat(np.arange(N), k)
    for z in dup:  # duplicate the previous column of the same sample -> summed by the COO mat-vec
        if z > 0 and row_of[z] == row_of[z - 1]:
            cols[z] = cols[z - 1]
    vals = rng.standard_normal(Z).astype(np.float32)
    offset = rng.standard_normal(N).astype(np.float32)
    y = (rng.random(N) < 0.5).astype(np.float32)
    weight = (0.25 + 2.0 * rng.random(N)).astype(np.float32) if random_weights else None
    return RawBatch(
        ent_row_ptr=np.concatenate([[0], np.cumsum(n)]),
        row_nnz_ptr=np.concatenate([[0], np.cumsum(k)]),
        col_global=cols, val=vals, y=y, offset=offset, weight=weight,
        uid=np.arange(N, dtype=np.int64), entity_ids=[f"r{i}" for i in range(E)])


def make_movielens_like(E, kind="per_user", seed=100, mean_n=None):
    """MovieLens-shaped entities (see module docstring). kind: per_user (D=20) | per_movie (D=24)."""
    rng = np.random.default_rng(seed)
    if kind == "per_user":
        mean_n = mean_n or 85
        n = np.maximum(1, rng.geometric(1.0 / mean_n, size=E)).astype(np.int64) + 15
        N = int(n.sum())
        n_genre = rng.integers(1, 7, size=N)
        k = n_genre + 1
        Z = int(k.sum())
        cols = np.empty(Z, np.int64)
        vals = np.empty(Z, np.float32)
        ptr = np.concatenate([[0], np.cumsum(k)])
        for i in range(N):
            g = np.sort(rng.choice(19, size=n_genre[i], replace=False))
            cols[ptr[i]:ptr[i] + n_genre[i]] = g
            vals[ptr[i]:ptr[i] + n_genre[i]] = 1.0
            cols[ptr[i + 1] - 1] = 19
            vals[ptr[i + 1] - 1] = np.float32(rng.integers(1930, 1999) / 2000.0)
        D = 20
    elif kind == "per_movie":
        mean_n = mean_n or 48
        n = np.maximum(1, rng.geometric(1.0 / mean_n, size=E)).astype(np.int64)
        N = int(n.sum())
        k = np.full(N, 3, np.int64)
        cols = np.stack([np.zeros(N, np.int64), 1 + rng.integers(0, 2, size=N),
                         3 + rng.integers(0, 21, size=N)], axis=1).reshape(-1)
        vals = np.stack([(rng.integers(7, 74, size=N) / 100.0), np.ones(N), np.ones(N)],
                        axis=1).astype(np.float32).reshape(-1)
        ptr = np.arange(N + 1, dtype=np.int64) * 3
        D = 24
    else:
        raise ValueError(kind)
    offset = (0.8 * rng.standard_normal(N)).astype(np.float32)
    w_star = 0.6 * rng.standard_normal(D)
    b_e = 0.7 * rng.standard_normal(E)
    row_of = np.repeat(np.arange(N), k)
    logit = np.bincount(row_of, weights=vals.astype(np.float64) * w_star[cols], minlength=N)
    logit = logit + np.repeat(b_e, n) + offset
    y = (rng.random(N) < 1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
    return RawBatch(ent_row_ptr=np.concatenate([[0], np.cumsum(n)]), row_nnz_ptr=ptr, col_global=cols,
                    val=vals, y=y, offset=offset, weight=None, uid=np.arange(N, dtype=np.int64),
                    entity_ids=[str(i + 1) for i in range(E)])


def algorithmic_bytes(batch_n, batch_nnz, batch_p, warm=False, var=False):
    """B(e) of SURVEY.md §8(d): 8*nnz + 16*n + 8*p*(1+warm+var) + 32, summed over the given entities."""
    n = np.asarray(batch_n, np.float64)
    z = np.asarray(batch_nnz, np.float64)
    p = np.asarray(batch_p, np.float64)
    return float((8.0 * z + 16.0 * n + 8.0 * p * (1 + int(warm) + int(var)) + 32.0).sum())
