"""Seeded synthetic entity-grouped data of the shapes BASELINE.json / SURVEY.md §8(d) name.

* make_survey_batch — the generator SURVEY.md §8(d) states, to the letter (C2, and the C5 mean shape): k DISTINCT uniform
  columns per sample in draw order, no hidden per-entity bias. This is what bench.py measures C2 on.
* make_batch — the stratified generator the committed parity fixtures were produced with (one column per stratum of the
  global feature space, so distinct and ascending by construction; a hidden per-entity bias b_e ~ 0.5 N(0,1)); also the
  Zipf-sized exploration shape. Kept as it is so that tests/golden regenerates bit for bit.
* make_ragged_batch — adversarially ragged entities (empty samples, repeated columns, weights).
* make_movielens_like / make_movielens_20m — MovieLens-shaped entities at ML-100K and ML-20M entity sizes (C1 / C3).
  There is no network in the build or GPU containers, so MovieLens itself cannot be downloaded; the feature bags are those
  of the reference's preprocessing script (scripts/download_process_movieLens_data.py:306-346,384-387): per_user = the
  rated movie's 1-6 genre flags (value 1.0) out of 19 + release_date/2000 (D = 20); per_movie = the rating user's
  age/100 + gender one-hot + occupation one-hot (k = 3, D = 24).
* make_c5_share_device — C5's per-GPU share (millions of Zipf-sized entities, up to 2^20 non-zeros each), generated in HBM
  by torch's counter-based (Philox) generator because a host generator would take minutes at a billion non-zeros.

In all of them: fp32 values, an fp32 offset (the fixed-effect score) and a label y ~ Bernoulli(sigmoid(x . w* [+ b_e] +
offset)) with a hidden global w*. Sample weights are 1 unless asked.
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
    if shape == "c5zipf":
        # SURVEY.md §8(d)'s C5 sizes to the letter (c5_entity_samples below): P(nnz >= x) ~ x^-1.2 on [k, 2^20], mean rescaled; the
        # caller's mean_n is the mean number of samples (mean nnz / k with k = 8)
        return c5_entity_samples(rng, E, mean_nnz=8 * mean_n, k=8)
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


# MovieLens-20M as the public dataset card states it (BASELINE.json configs[2]): 138 493 users with at least 20 ratings each,
# 26 744 rated movies, 20 000 263 ratings; the reference keeps a random 80 % of the rows for training
# (scripts/download_process_movieLens_data.py:154-165,408).
ML20M_USERS, ML20M_MOVIES, ML20M_RATINGS, ML20M_TRAIN = 138_493, 26_744, 20_000_263, 0.8
ML20M_MAX_PER_USER, ML20M_MAX_PER_MOVIE = 9_254, 67_310


def _ml20m_user_counts(rng, U):
    """Ratings per user: 20 + a log-normal tail (median 68, mean 144, clipped at the most active user's 9 254)."""
    extra = np.exp(rng.normal(np.log(48.0), 1.38, size=U))
    n = np.minimum(20 + np.floor(extra), ML20M_MAX_PER_USER).astype(np.int64)
    return n


def _ml20m_movie_counts(M, total):
    """Ratings per movie by popularity rank: log-count interpolated through anchor points of the public dataset's long tail
    (67 k ratings for the most rated title, ~1 000 at rank 3 900, ~100 at rank 8 500, ~10 at rank 15 500, a single rating
    for the last 4 000 titles), the body then scaled so that the counts add up to `total` rows."""
    rank = np.array([0, 10, 100, 500, 1000, 3900, 8500, 15500, 19500, 22800, M - 1], np.float64) * (M - 1) / (ML20M_MOVIES - 1)
    cnt = np.array([ML20M_MAX_PER_MOVIE, 50000, 30000, 12000, 7000, 1000, 100, 10, 3, 1.4, 1.0])
    n = np.exp(np.interp(np.arange(M, dtype=np.float64), rank, np.log(cnt)))
    body = np.arange(M) > 0
    for _ in range(8):   # the head and the floor of one rating stay where they are
        scaled = np.maximum(1.0, n * np.where(body, (total - n[0]) / n[body].sum(), 1.0))
        n = np.minimum(scaled, ML20M_MAX_PER_MOVIE)
    return np.maximum(1, np.round(n)).astype(np.int64)


def _ragged_from_padded(table, count, pick):
    """Rows `pick` of a padded [R, W] table, each cut to count[pick] entries, flattened in row order."""
    sub = table[pick]
    keep = np.arange(table.shape[1])[None, :] < count[pick][:, None]
    return sub[keep]


def make_movielens_20m(kind="per_user", seed=200, entities=None, with_uid=False):
    """MovieLens-20M-sized random-effect entities (BASELINE.json configs[2], C3): the count distributions of the real dataset
    (see the constants above) with the reference's feature bags. kind = per_user: one entity per user, a sample per rated
    movie (k = 2..7, D = 20, up to ~7.4 k training rows per user); per_movie: one entity per movie, a sample per rating
    user (k = 3, D = 24; the most rated title keeps ~54 k training rows, a third of the catalogue one or two): tall and
    skinny, n >> p. `entities` keeps a random subset of that many entities (tests at reduced size); entity order is random,
    as a hash partition's is."""
    rng = np.random.default_rng(seed)
    U, M = ML20M_USERS, ML20M_MOVIES
    user_full = _ml20m_user_counts(rng, U)
    movie_full = _ml20m_movie_counts(M, ML20M_RATINGS)
    if kind == "per_user":
        n = np.maximum(1, rng.binomial(user_full, ML20M_TRAIN)).astype(np.int64)
        n = n[rng.permutation(U)]
        if entities is not None:
            n = n[:int(entities)]
        E, N = n.size, int(n.sum())
        # the movie catalogue: 1-6 genres (sorted indices 0..18) + release year
        ng = rng.choice(np.arange(1, 7), size=M, p=[0.36, 0.33, 0.20, 0.08, 0.025, 0.005])
        genre_pop = np.array([0.2, 3.5, 2.3, 1.0, 1.1, 8.4, 2.9, 2.5, 13.3, 1.4, 0.3, 2.6, 1.0, 1.5, 4.1, 1.7, 4.2, 1.2, 0.7])
        keys = rng.gumbel(size=(M, 19)) + np.log(genre_pop)[None, :]          # Gumbel top-k = sampling without replacement
        top = np.argsort(-keys, axis=1)[:, :6]
        top[np.arange(6)[None, :] >= ng[:, None]] = 99
        top = np.sort(top, axis=1)                                             # ascending genre index, padding last
        table = np.full((M, 7), 19, np.int8)
        table[:, :6] = np.where(top == 99, 19, top)
        # the release-date column sits right after the movie's genres
        ctab = np.full((M, 7), 19, np.int8)
        vtab = np.zeros((M, 7), np.float32)
        year = (rng.integers(1915, 2016, size=M) / 2000.0).astype(np.float32)
        for j in range(6):
            is_genre = j < ng
            ctab[:, j] = np.where(is_genre, table[:, j], 19)
            vtab[:, j] = np.where(is_genre, 1.0, year)
        vtab[:, 6] = year
        cnt = (ng + 1).astype(np.int64)
        cdf = np.cumsum(movie_full / movie_full.sum())
        pick = np.minimum(np.searchsorted(cdf, rng.random(N)), M - 1)          # a popular movie is rated by many users
        cols = _ragged_from_padded(ctab, cnt, pick).astype(np.int64)
        vals = _ragged_from_padded(vtab, cnt, pick)
        k = cnt[pick]
        D = 20
    elif kind == "per_movie":
        n = np.maximum(1, rng.binomial(movie_full, ML20M_TRAIN)).astype(np.int64)
        n = n[rng.permutation(M)]
        if entities is not None:
            n = n[:int(entities)]
        E, N = n.size, int(n.sum())
        age = (rng.integers(7, 74, size=U) / 100.0).astype(np.float32)
        gender = (1 + (rng.random(U) < 0.29)).astype(np.int64)
        occupation = 3 + rng.integers(0, 21, size=U)
        cdf = np.cumsum(user_full / user_full.sum())
        pick = np.minimum(np.searchsorted(cdf, rng.random(N)), U - 1)          # an active user rates many movies
        cols = np.stack([np.zeros(N, np.int64), gender[pick], occupation[pick]], axis=1).reshape(-1)
        vals = np.stack([age[pick], np.ones(N, np.float32), np.ones(N, np.float32)], axis=1).reshape(-1)
        k = np.full(N, 3, np.int64)
        D = 24
    else:
        raise ValueError(kind)
    ptr = np.concatenate([[0], np.cumsum(k)])
    offset = (0.8 * rng.standard_normal(N)).astype(np.float32)
    w_star = 0.6 * rng.standard_normal(D)
    b_e = 0.7 * rng.standard_normal(E)
    row_of = np.repeat(np.arange(N), k)
    logit = np.bincount(row_of, weights=vals.astype(np.float64) * w_star[cols], minlength=N)
    logit = logit + np.repeat(b_e, n) + offset
    y = (rng.random(N) < 1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
    return RawBatch(ent_row_ptr=np.concatenate([[0], np.cumsum(n)]), row_nnz_ptr=ptr, col_global=cols, val=vals, y=y, offset=offset,
                    weight=None, uid=np.arange(N, dtype=np.int64) if with_uid else None, entity_ids=[str(i + 1) for i in range(E)],
                    trusted=True)


def c5_entity_samples(rng, E, mean_nnz=256, k=8, max_nnz=1 << 20):
    """Samples per entity of SURVEY.md §8(d)'s C5: nnz_e Zipf-like with P(nnz >= x) ~ x^-1.2, truncated to [k, max_nnz] and
    rescaled to the requested mean; n_e = nnz_e / k."""
    raw = (1.0 - rng.random(E)) ** (-1.0 / 1.2)
    cap, want = max_nnz // k, mean_nnz / k
    c = want / raw.mean()
    for _ in range(30):    # the cap removes mass from the tail: solve for the scale that restores the mean
        n = np.clip(np.floor(raw * c), 1, cap)
        c_next = c * (want / n.mean())
        if c_next == c:    # a fixed point (reached after a handful of rounds): the remaining rounds would not change anything
            break
        c = c_next
    return np.clip(np.floor(raw * c), 1, cap).astype(np.int64)


def _c5_rows(g, dev, rows, k, D, w_star, cols, vals, offset, y, at, chunk_rows=1 << 24):
    """`rows` samples of SURVEY.md §8(d)'s C5 written at [at, at + rows) of the output arrays, drawn from generator g."""
    import torch
    for r0 in range(0, rows, chunk_rows):
        r1 = min(rows, r0 + chunk_rows)
        c = torch.randint(0, D, (r1 - r0, k), generator=g, device=dev, dtype=torch.int64)
        while True:    # re-draw the samples that drew a column twice
            s, _ = torch.sort(c, dim=1)
            bad = torch.nonzero((s[:, 1:] == s[:, :-1]).any(dim=1)).reshape(-1)
            if bad.numel() == 0:
                break
            c[bad] = torch.randint(0, D, (bad.numel(), k), generator=g, device=dev, dtype=torch.int64)
        v = torch.randn((r1 - r0, k), generator=g, device=dev, dtype=torch.float32)
        o = torch.randn(r1 - r0, generator=g, device=dev, dtype=torch.float32)
        logit = (v.double() * w_star[c]).sum(dim=1) + o.double()
        cols[at + r0:at + r1], vals[at + r0:at + r1], offset[at + r0:at + r1] = c, v, o
        y[at + r0:at + r1] = (torch.rand(r1 - r0, generator=g, device=dev, dtype=torch.float64) < torch.sigmoid(logit)).float()
        del c, v, o, logit, s, bad


def make_c5_share_device(device, E, seed=C5_SEED, mean_nnz=256, k=8, D=65536, max_nnz=1 << 20, chunk_rows=1 << 24):
    """C5's per-GPU share (BASELINE.json configs[4]) generated in HBM: entity sizes on the host (numpy, seeded),
    everything per sample on the device with torch's Philox generator — k distinct uniform columns of D per sample, values ~
    N(0,1) fp32, offset ~ N(0,1) fp32, y ~ Bernoulli(sigmoid(x . w* + offset)), weight 1. Returns (raw, n) with raw the dict
    of device tensors REDeviceSolver.pack takes and n the host array of samples per entity. Uses torch for plumbing only
    (random numbers and element-wise arithmetic of the *input*); nothing of the solve."""
    import torch
    rng = np.random.default_rng(seed)
    n = c5_entity_samples(rng, E, mean_nnz, k, max_nnz)
    N = int(n.sum())
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    w_star = 0.5 * torch.randn(D, generator=g, device=dev, dtype=torch.float64)
    cols = torch.empty((N, k), dtype=torch.int64, device=dev)
    vals = torch.empty((N, k), dtype=torch.float32, device=dev)
    offset = torch.empty(N, dtype=torch.float32, device=dev)
    y = torch.empty(N, dtype=torch.float32, device=dev)
    _c5_rows(g, dev, N, k, D, w_star, cols, vals, offset, y, 0, chunk_rows)
    ent_row_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(n)])).to(dev)
    raw = dict(E=int(E), N=N, Z=N * k, ent_row_ptr=ent_row_ptr, row_nnz_ptr=torch.arange(N + 1, dtype=torch.int64, device=dev) * k,
               col_global=cols.reshape(-1), val=vals.reshape(-1), y=y, offset=offset, weight=None)
    return raw, n


def rank_partitions(num_partitions, world, rank):
    """The partitions worker `rank` of `world` trains: partitions[rank::world] of the partition list 0..P-1
    (drivers/random_effect_driver.py:60-68)."""
    return list(range(int(num_partitions)))[int(rank)::int(world)]


def population_share(entity_ids, num_partitions, world, rank, partition_ids=None):
    """One population split the reference's way: entity -> partition by the Java hash of its decimal id (PartitionUtils.scala:31-37,
    contract B4), partition -> worker by partitions[rank::world]. -> (indices of this rank's entities, ordered by partition and,
    inside a partition, by their order in `entity_ids`; their partition ids). `partition_ids` (from the device routine
    gdmix_java_partition_ids_i64) may be handed in; the host routine of partitioner.py is the default."""
    ids = np.asarray(entity_ids, np.int64)
    if partition_ids is None:
        from .partitioner import java_partition_ids_int64
        partition_ids = java_partition_ids_int64(ids, num_partitions)
    pid = np.asarray(partition_ids, np.int64)
    mine = np.zeros(int(num_partitions), bool)
    mine[rank_partitions(num_partitions, world, rank)] = True
    own = np.flatnonzero(mine[pid])
    own = own[np.argsort(pid[own], kind="stable")]
    return own, pid[own]


def make_c5_population_share(device, E_total, world, rank, num_partitions=1024, seed=C5_SEED, mean_nnz=256, k=8, D=65536,
                             max_nnz=1 << 20, partition_ids_fn=None):
    """Rank `rank`'s share of ONE C5 population of E_total entities (BASELINE.json configs[4]; SURVEY.md §8(d): "partition = Java
    hash (B4) of the decimal entity id into 1 024 partitions"), generated in HBM. The population does not depend on the number
    of workers: entity e has id e and n_e samples from one seeded host draw over all E_total entities; its partition is the
    Java hash of str(e); the samples of partition K come from a Philox stream seeded by (seed, K), entities of K in id order.
    Worker r holds partitions[r::world], concatenated in partition order. -> (raw, n, ids, pid): the dict of device tensors
    REDeviceSolver.pack takes, and per entity of the share (host arrays) its samples, global id and partition."""
    import torch
    rng = np.random.default_rng(seed)
    n_all = c5_entity_samples(rng, int(E_total), mean_nnz, k, max_nnz)
    ids_all = np.arange(int(E_total), dtype=np.int64)
    pid_all = None if partition_ids_fn is None else np.asarray(partition_ids_fn(ids_all, num_partitions))
    own, pid = population_share(ids_all, num_partitions, world, rank, pid_all)
    n = n_all[own]
    N = int(n.sum())
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    w_star = 0.5 * torch.randn(D, generator=g, device=dev, dtype=torch.float64)
    cols = torch.empty((N, k), dtype=torch.int64, device=dev)
    vals = torch.empty((N, k), dtype=torch.float32, device=dev)
    offset = torch.empty(N, dtype=torch.float32, device=dev)
    y = torch.empty(N, dtype=torch.float32, device=dev)
    at = 0
    rows_of = np.bincount(pid, weights=n, minlength=int(num_partitions)).astype(np.int64)
    for K in rank_partitions(num_partitions, world, rank):
        rows = int(rows_of[K])
        if rows:
            g.manual_seed(int(seed) * 1_000_003 + 1 + K)
            _c5_rows(g, dev, rows, k, D, w_star, cols, vals, offset, y, at)
            at += rows
    ent_row_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(n)])).to(dev)
    raw = dict(E=int(own.size), N=N, Z=N * k, ent_row_ptr=ent_row_ptr, row_nnz_ptr=torch.arange(N + 1, dtype=torch.int64, device=dev) * k,
               col_global=cols.reshape(-1), val=vals.reshape(-1), y=y, offset=offset, weight=None)
    return raw, n, own, pid


class C5Population:
    """ONE C5 population (BASELINE.json configs[4]), partition by partition: the same entities, sizes, partitions
    and per-partition Philox streams as make_c5_population_share, but a partition is generated on its own — what a worker of the
    product path holds at a time (drivers/random_effect_driver.py:60-68: one partition per round). 100 M entities: the sizes are
    one seeded host draw (800 MB), the partition of every id the Java hash of its decimal form (partition_ids_fn: the device
    routine gdmix_java_partition_ids_i64; default the host routine of partitioner.py)."""

    def __init__(self, E_total, num_partitions=1024, seed=C5_SEED, mean_nnz=256, k=8, D=65536, max_nnz=1 << 20, partition_ids_fn=None):
        self.E_total, self.P, self.seed, self.k, self.D = int(E_total), int(num_partitions), int(seed), int(k), int(D)
        self.n_all = c5_entity_samples(np.random.default_rng(seed), self.E_total, mean_nnz, k, max_nnz)
        ids = np.arange(self.E_total, dtype=np.int64)
        if partition_ids_fn is None:
            from .partitioner import java_partition_ids_int64
            pid = java_partition_ids_int64(ids, self.P)
        else:
            pid = np.asarray(partition_ids_fn(ids, self.P))
        self.pid_all = pid.astype(np.int32)
        self.order = np.argsort(self.pid_all, kind="stable").astype(np.int64)        # ids by partition, in id order inside one
        self.part_ptr = np.concatenate([[0], np.cumsum(np.bincount(self.pid_all, minlength=self.P))]).astype(np.int64)
        self._w_star = {}

    def ids_of(self, K):
        return self.order[self.part_ptr[K]:self.part_ptr[K + 1]]

    def partition(self, K, device):
        """-> (raw, n, ids): partition K's raw batch in HBM (the dict REDeviceSolver.pack takes), samples and global id per entity."""
        import torch
        dev = torch.device(device)
        ids = self.ids_of(K)
        n = self.n_all[ids]
        N, k = int(n.sum()), self.k
        g = torch.Generator(device=dev)
        if dev not in self._w_star:
            g.manual_seed(self.seed)
            self._w_star[dev] = 0.5 * torch.randn(self.D, generator=g, device=dev, dtype=torch.float64)
        cols = torch.empty((N, k), dtype=torch.int64, device=dev)
        vals = torch.empty((N, k), dtype=torch.float32, device=dev)
        offset = torch.empty(N, dtype=torch.float32, device=dev)
        y = torch.empty(N, dtype=torch.float32, device=dev)
        if N:
            g.manual_seed(self.seed * 1_000_003 + 1 + int(K))
            _c5_rows(g, dev, N, k, self.D, self._w_star[dev], cols, vals, offset, y, 0)
        raw = dict(E=int(ids.size), N=N, Z=N * k, ent_row_ptr=torch.from_numpy(np.concatenate([[0], np.cumsum(n)])).to(dev),
                   row_nnz_ptr=torch.arange(N + 1, dtype=torch.int64, device=dev) * k, col_global=cols.reshape(-1), val=vals.reshape(-1),
                   y=y, offset=offset, weight=None)
        return raw, n, ids


def subset_raw(raw, n, ents):
    """Entities `ents` (any order) of a device raw batch with a constant number of non-zeros per sample, as a raw batch of their own
    in HBM — what the re-balancer's exchange makes of a partition, without the ranks (bench_strong's projection of a plan)."""
    import torch
    ents = np.asarray(ents, np.int64)
    ptr = np.concatenate([[0], np.cumsum(n)])
    k = raw["Z"] // max(1, raw["N"])
    ne = n[ents]
    out_start = np.cumsum(ne) - ne
    N = int(ne.sum())
    dev = raw["y"].device
    rows = torch.from_numpy(np.arange(N, dtype=np.int64) - np.repeat(out_start, ne) + np.repeat(ptr[ents], ne)).to(dev)
    return dict(E=int(ents.size), N=N, Z=N * k, ent_row_ptr=torch.from_numpy(np.concatenate([[0], np.cumsum(ne)])).to(dev),
                row_nnz_ptr=torch.arange(N + 1, dtype=torch.int64, device=dev) * k,
                col_global=raw["col_global"].reshape(-1, k)[rows].reshape(-1), val=raw["val"].reshape(-1, k)[rows].reshape(-1),
                y=raw["y"][rows], offset=raw["offset"][rows], weight=None), ne


def concat_raw(parts):
    """Raw batches (constant non-zeros per sample, no weights) one after another as one raw batch in HBM."""
    import torch
    parts = [p for p in parts if p["E"] > 0]
    if len(parts) == 1:
        return parts[0]
    k = parts[0]["Z"] // max(1, parts[0]["N"])
    N = sum(p["N"] for p in parts)
    dev = parts[0]["y"].device
    ptrs, base = [torch.zeros(1, dtype=torch.int64, device=dev)], 0
    for p in parts:
        ptrs.append(p["ent_row_ptr"][1:] + base)
        base += p["N"]
    return dict(E=sum(p["E"] for p in parts), N=N, Z=N * k, ent_row_ptr=torch.cat(ptrs),
                row_nnz_ptr=torch.arange(N + 1, dtype=torch.int64, device=dev) * k, col_global=torch.cat([p["col_global"] for p in parts]),
                val=torch.cat([p["val"] for p in parts]), y=torch.cat([p["y"] for p in parts]), offset=torch.cat([p["offset"] for p in parts]),
                weight=None)


def device_entities_to_host(raw, n, ents):
    """The entities `ents` of a device raw batch with a constant number of non-zeros per sample (make_c5_share_device) as a
    host RawBatch (for comparing a sample with the CPU checker)."""
    import torch
    ents = np.asarray(ents, np.int64)
    ptr = np.concatenate([[0], np.cumsum(n)])
    k = raw["Z"] // max(1, raw["N"])
    ne = n[ents]
    out_start = np.cumsum(ne) - ne
    rows = torch.from_numpy(np.arange(int(ne.sum()), dtype=np.int64) - np.repeat(out_start, ne) + np.repeat(ptr[ents], ne)).to(raw["y"].device)
    N = int(ne.sum())
    g = lambda t: t[rows].cpu().numpy()
    return RawBatch(ent_row_ptr=np.concatenate([[0], np.cumsum(ne)]), row_nnz_ptr=np.arange(N + 1, dtype=np.int64) * k,
                    col_global=raw["col_global"].reshape(-1, k)[rows].cpu().numpy().reshape(-1),
                    val=raw["val"].reshape(-1, k)[rows].cpu().numpy().reshape(-1), y=g(raw["y"]), offset=g(raw["offset"]),
                    weight=None, entity_ids=[str(int(e)) for e in ents])


def algorithmic_bytes(batch_n, batch_nnz, batch_p, warm=False, var=False):
    """B(e) of SURVEY.md §8(d): 8*nnz + 16*n + 8*p*(1+warm+var) + 32, summed over the given entities."""
    n = np.asarray(batch_n, np.float64)
    z = np.asarray(batch_nnz, np.float64)
    p = np.asarray(batch_p, np.float64)
    return float((8.0 * z + 16.0 * n + 8.0 * p * (1 + int(warm) + int(var)) + 32.0).sum())
