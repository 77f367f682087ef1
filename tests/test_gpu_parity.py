"""Parity of the HIP path (through the C ABI) against (1) the golden vectors generated from the
reference itself and (2) the CPU oracle on the same seeded inputs. Runs on the MI355X box only.

Bars (BASELINE.json north_star): coefficients within 1e-5 relative error of the reference L-BFGS
result on well-posed entities — asserted here two orders tighter (1e-7) — and bit-exact integer work
(np.unique / local indices / CSC order / partition ids).
"""
import os

import numpy as np
import pytest

from helpers import (check_d_class, fixture_names, load_fixture, opts_kwargs, parity_mask, per_entity_rel_err, well_posed_mask)
from gdmix_amd import synthetic
from gdmix_amd.solver import SolverOptions
from oracle import oracle

pytestmark = pytest.mark.gpu

REL_TOL_DEVICE = 1e-7     # vs reference fixtures / oracle, well-posed entities (north star: 1e-5)


def _expected_csc(pk, val):
    """CSC copy the pack kernel must produce: non-zeros sorted by (local col, position)."""
    E = pk["E"]
    col_ptr = []
    csc_row = np.zeros(pk["Z"], np.int32)
    csc_val = np.zeros(pk["Z"], np.float32)
    for e in range(E):
        z0, z1 = pk["ent_nnz_ptr"][e], pk["ent_nnz_ptr"][e + 1]
        r0, r1 = pk["ent_row_ptr"][e], pk["ent_row_ptr"][e + 1]
        f0, f1 = pk["ent_feat_ptr"][e], pk["ent_feat_ptr"][e + 1]
        rp = pk["row_ptr"][r0 + e: r1 + e + 1]
        rows = np.repeat(np.arange(r1 - r0), np.diff(rp))
        cols = pk["csr_col"][z0:z1]
        order = np.argsort(cols, kind="stable")
        csc_row[z0:z1] = rows[order]
        csc_val[z0:z1] = val[z0:z1][order]
        col_ptr.append(np.concatenate([[0], np.cumsum(np.bincount(cols, minlength=f1 - f0))]).astype(np.int32))
    return col_ptr, csc_row, csc_val


def _check_pack(packed, pk, val):
    assert packed.D == pk["D"]
    assert np.array_equal(packed.ent_feat_ptr().cpu().numpy(), pk["ent_feat_ptr"])
    assert np.array_equal(packed.ent_nnz_ptr().cpu().numpy(), pk["ent_nnz_ptr"])
    assert np.array_equal(packed.unique_global().cpu().numpy(), pk["unique_global"])
    assert np.array_equal(packed.csr_col().cpu().numpy(), pk["csr_col"])
    assert np.array_equal(packed.row_ptr().cpu().numpy(), pk["row_ptr"])
    col_ptr, csc_row, csc_val = _expected_csc(pk, val)
    got_cp = packed.col_ptr().cpu().numpy()   # d_e + 1 entries at ent_nnz_ptr[e] + e
    for e, want in enumerate(col_ptr):
        z0 = int(pk["ent_nnz_ptr"][e]) + e
        assert np.array_equal(got_cp[z0:z0 + want.size], want)
    assert np.array_equal(packed.csc_row().cpu().numpy(), csc_row)
    assert np.array_equal(packed.csc_val().cpu().numpy(), csc_val)


def _solve_and_compare(device_solver, name, lds_limit=65536, kernel_mask=7, giant_nnz=16777216, team_nnz=16384, tall_min_n=0,
                       tall_split_n=None, tall_team_n=None, tall_mid_n=None):
    """tall_min_n: 0 keeps the tall kernel out of the way of the routing under test; None = the library's default."""
    b, opts, exp, _ = load_fixture(name)
    kw = opts_kwargs(opts)
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    packed = device_solver.pack(b, has_intercept=kw["has_intercept"])
    _check_pack(packed, pk, b.val)
    th0 = exp["theta0"] if np.any(exp["theta0"]) else None
    device_solver.set_wave_lds_limit(lds_limit)
    device_solver.set_kernel_mask(kernel_mask)
    device_solver.set_giant_nnz(giant_nnz)
    device_solver.set_team_nnz(team_nnz)
    device_solver.set_tall_min_n(device_solver.TALL_MIN_N_DEFAULT if tall_min_n is None else tall_min_n)
    if tall_split_n is not None:
        device_solver.set_tall_split_n(tall_split_n)
    if tall_team_n is not None:
        device_solver.set_tall_team_n(tall_team_n)
    if tall_mid_n is not None:
        device_solver.set_tall_mid_n(tall_mid_n)
    try:
        res = device_solver.solve(packed, SolverOptions(**kw), theta0=th0).to_host()
    finally:
        device_solver.set_tall_mid_n(0)
        device_solver.set_tall_split_n(0)
        device_solver.set_tall_team_n(device_solver.TALL_TEAM_N_DEFAULT)
        device_solver.set_wave_lds_limit(65536)
        device_solver.set_kernel_mask(7)
        device_solver.set_giant_nnz(16777216)
        device_solver.set_team_nnz(16384)
        device_solver.set_tall_min_n(device_solver.TALL_MIN_N_DEFAULT)
    ref = oracle.solve(pk, b.val, b.y, b.offset, b.weight, oracle.make_opts(**kw), theta0=th0)
    coef_ptr = packed.coef_ptr_host()
    wp = parity_mask(b, opts, exp)
    assert np.all(res["status"] >= 0)
    # (1) against the reference's own numbers
    err = per_entity_rel_err(res["theta"], exp["theta"], coef_ptr)
    assert err[wp].max() <= REL_TOL_DEVICE, f"{name}: theta rel err vs reference {err[wp].max():.3e}"
    assert np.array_equal(res["nit"][wp], exp["nit"][wp]), name
    # nfev is scipy's funcalls on both sides: ScalarFunction serves a trial point equal to the previously evaluated one from
    # its cache without counting it (a step too small to move x; the extreme entities of the exit_* sets)
    assert np.array_equal(res["nfev"][wp], exp["nfev"][wp]), name
    assert np.array_equal(res["nfev"][wp], ref["nfev"][wp]), name
    assert np.array_equal(res["status"][wp], exp["status"][wp]), name
    fv = wp & (exp["status"] != 4)   # f after ABNORMAL: scipy's driver reports the last trial's, see test_oracle_golden.py
    np.testing.assert_allclose(res["fval"][fv], exp["fval"][fv], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(res["fval"][wp], ref["fval"][wp], rtol=1e-9, atol=1e-13)
    # thresholded coefficients: same zero pattern, same values
    m = np.zeros(coef_ptr[-1], bool)
    for e in np.flatnonzero(wp):
        m[coef_ptr[e]:coef_ptr[e + 1]] = True
    assert np.array_equal((res["theta_thr"] == 0)[m], (exp["theta_thr"] == 0)[m]), name
    # (2) against the oracle on the same inputs
    err_o = per_entity_rel_err(res["theta"], ref["theta"], coef_ptr)
    assert err_o[wp].max() <= REL_TOL_DEVICE
    if kw["variance_mode"] in (1, 2):   # SIMPLE: same sums; FULL: Cholesky here vs LU (np.linalg.inv) there
        np.testing.assert_allclose(res["variance"], exp["variance"], rtol=1e-7)
    # degenerate entities: invariants only — the whole class-D contract of SURVEY.md §8(d)
    dg = ~wp
    if dg.any():
        conv = dg & (res["status"] == 0)
        assert np.all(res["gnorm"][conv] <= 1e-5)
    check_d_class(b, opts, exp, res, name)
    return dict(device_solver.class_counts(packed)), np.diff(coef_ptr), kw


@pytest.mark.parametrize("name", fixture_names())
def test_default_routing_matches_reference_fixture(device_solver, name):
    """Default routing: group kernels by size class, the tall kernel for p <= 64 with many samples, team kernels above."""
    _solve_and_compare(device_solver, name, tall_min_n=None)


@pytest.mark.parametrize("name", fixture_names())
def test_tall_kernel_matches_reference_fixture(device_solver, name):
    """Every entity with at most 64 coefficients (whatever its sample count: threshold 1) through the tall kernels — one
    workgroup per entity: the lean variant for those that fit a twelfth of a CU's LDS, one-wavefront workgroups for the others
    below 4 096 samples, eight wavefronts above; entities of more coefficients keep their default routing."""
    counts, p, kw = _solve_and_compare(device_solver, name, tall_min_n=1)
    want = int((p <= 64).sum()) if kw["m"] <= 10 else 0
    got = (counts["re_solve_tall_kernel<8> p<=64"] + counts["re_solve_tall_kernel<1> p<=64"] + counts["re_solve_tall_kernel<1> lean p<=64"]
           + counts["re_solve_tall_team_kernel<8> x4 p<=64"]      # (the one title above 8 192 samples of ml20m_per_movie_tall gets a team: round 4)
           + counts["re_solve_tall_kernel<4> p<=64"])             # (round 6: the mid class; none unless switched on)
    assert got == want, (got, want)


@pytest.mark.parametrize("name", ["ref_fixture_l2_0.1", "ref_dataset1", "c2_shipped_cfg", "c2_weights", "c2_no_intercept", "c2_maxiter1", "c2_m3",
                                  "ragged", "ragged_variance_simple", "ml_per_user", "ml_per_movie", "warm_stage2", "tiny_entities_regbias",
                                  "exit_factr_1e-7", "exit_hard_02", "exit_extreme_00", "exit_extreme_02", "ml20m_per_movie_tall", "ml20m_per_user_tall"])
def test_tall_mid_kernel_matches_reference_fixture(device_solver, name):
    """Round 6: the MID tall class (re_solve_tall_kernel<4>: four wavefronts per entity, two workgroups per CU, half a CU's LDS each).
    Every entity with at most 64 coefficients that is not lean and has at least 8 samples through it (threshold 8; the split pinned at
    its default so that nothing above moves away): same fixtures, same tolerances, same iteration counts."""
    b = load_fixture(name)[0]
    counts, p, kw = _solve_and_compare(device_solver, name, tall_min_n=1, tall_split_n=4096, tall_team_n=0, tall_mid_n=8)
    n = b.ent_n()
    tall = (p <= 64) if kw["m"] <= 10 else np.zeros_like(p, bool)
    mid = counts["re_solve_tall_kernel<4> p<=64"]
    assert mid + counts["re_solve_tall_kernel<1> lean p<=64"] + counts["re_solve_tall_kernel<1> p<=64"] == int((tall & (n < 4096)).sum())
    assert counts["re_solve_tall_kernel<1> p<=64"] <= int((tall & (n < 8)).sum())      # what is left on one wavefront is below the threshold
    if name.startswith("ml20m"):
        assert mid > 0


@pytest.mark.parametrize("name", ["ref_fixture_l2_0.1", "ref_dataset1", "c2_shipped_cfg", "c2_weights", "c2_no_intercept", "c2_maxiter1", "c2_m3",
                                  "ragged", "ragged_variance_simple", "ml_per_user", "ml_per_movie", "warm_stage2", "tiny_entities_regbias",
                                  "exit_factr_1e-7", "exit_hard_02", "exit_extreme_00", "exit_extreme_02", "ml20m_per_movie_tall", "ml20m_per_user_tall"])
def test_tall_team_kernel_matches_reference_fixture(device_solver, name):
    """Round 4: FOUR workgroups on one entity (csrc/re_solve_tall.hip, re_solve_tall_team_kernel). Every entity with at most 64
    coefficients and at least 64 samples through it (split 1: all of them are eight-wavefront entities; team threshold -64: no limit on
    the class), the smaller ones through the one-workgroup kernel: same fixtures, same tolerances, same iteration counts."""
    b = load_fixture(name)[0]
    counts, p, kw = _solve_and_compare(device_solver, name, tall_min_n=1, tall_split_n=1, tall_team_n=-64)
    n = b.ent_n()
    want = int(((p <= 64) & (n >= 64)).sum()) if kw["m"] <= 10 else 0
    assert counts["re_solve_tall_team_kernel<8> x4 p<=64"] == want, (counts, want)
    assert counts["re_solve_tall_kernel<8> p<=64"] == (int(((p <= 64) & (n < 64)).sum()) if kw["m"] <= 10 else 0)
    if name.startswith("ml20m"):
        assert want > 0


VARIANT_FIXTURES = ["ref_fixture_l2_0.1", "ref_dataset1", "ref_dataset2", "c2_shipped_cfg", "c2_defaults", "c2_l2_1e-3",
                    "c2_large_offsets", "c2_weights", "c2_no_intercept", "c2_maxiter1", "c2_maxiter3_m2", "c2_m3",
                    "ragged", "ragged_variance_simple", "ragged_variance_full", "ref_dataset1_variance_full", "ml_per_user", "ml_per_movie", "c5_mean_shape", "zipf_tail",
                    "tiny_entities_regbias", "tiny_entities_shipped_cfg", "warm_stage2",
                    "exit_factr_1e-7", "exit_factr_1e-4_m3_weights", "exit_hard_02", "exit_hard_04", "exit_extreme_00", "exit_extreme_01",
                    "exit_extreme_02", "exit_extreme_03", "ml20m_per_movie_tall", "ml20m_per_user_tall"]


def test_default_routing_reaches_only_these_classes(device_solver):
    """Which size classes does the default routing use? BASELINE-shaped batches (C2, C5 / Zipf, MovieLens per-user and per-movie at
    ML-100K and ML-20M sizes), ragged and tiny ones, wide and tall ones, at m = 10 and (the LDS-resident wavefront kernel's reason
    to exist) at m = 12: group kernels, the tall kernels, the team kernels — and the wavefront kernel for m > 10 only. Round 1's
    register wavefront kernel had thirteen classes that none of this reached; they were removed in round 4."""
    from gdmix_amd.batch import concat
    from gdmix_amd.solver import NUM_CLASSES
    shapes = [synthetic.make_survey_batch(4000, 16, 4, 1024, seed=1), synthetic.make_batch(3000, 32, 8, 65536, seed=2, size_dist="zipf"),
              synthetic.make_movielens_like(600, "per_user", seed=3), synthetic.make_movielens_like(600, "per_movie", seed=4),
              synthetic.make_movielens_20m("per_user", seed=5, entities=500), synthetic.make_movielens_20m("per_movie", seed=6, entities=900),
              synthetic.make_ragged_batch(800, seed=7), synthetic.make_batch(300, 1, 2, 16, seed=8, size_dist="const"),
              synthetic.make_batch(30, 20, 128, 65536, seed=9, size_dist="const"), synthetic.make_batch(3, 9000, 2, 512, seed=10, size_dist="const"),
              synthetic.make_batch(4, 3000, 1, 400, seed=11, size_dist="const")]      # (n > every group kernel's sample cap, 64 < p <= 512)
    used = {10: set(), 12: set()}
    for m in (10, 12):
        for b in shapes:
            packed = device_solver.pack(b)
            res = device_solver.solve(packed, SolverOptions(l2=1.0, regularize_bias=False, m=m, max_iter=5))
            assert int((res.status < 0).sum().item()) == 0
            used[m] |= {name.split("<")[0].split(" ")[0] for name, c in device_solver.class_counts(packed) if c > 0}
    assert len(device_solver.class_counts(packed)) == NUM_CLASSES == 40
    # (the 900 MovieLens-20M movies include titles above 8 192 samples: they get a team of workgroups since round 4)
    assert used[10] == {"re_solve_grp_kernel", "re_solve_tall_kernel", "re_solve_tall_team_kernel", "re_solve_team_kernel"}, used[10]
    assert used[12] == {"re_solve_wave_kernel", "re_solve_team_kernel"}, used[12]      # m > 10: the LDS wavefront kernel and the two-loop block kernel


@pytest.mark.parametrize("name", VARIANT_FIXTURES)
def test_lds_wave_kernel_matches_reference_fixture(device_solver, name):
    _solve_and_compare(device_solver, name, kernel_mask=2)


@pytest.mark.parametrize("name", VARIANT_FIXTURES)
def test_quad_kernel_matches_reference_fixture(device_solver, name):
    _solve_and_compare(device_solver, name, kernel_mask=4)


@pytest.mark.parametrize("name", ["ref_fixture_l2_0.1", "c2_shipped_cfg", "c2_l2_1e-3", "ragged", "ml_per_user",
                                  "c5_mean_shape", "warm_stage2", "tiny_entities_regbias", "c2_no_intercept",
                                  "ragged_variance_simple", "c2_m3", "exit_factr_1e-4", "exit_hard_02", "exit_hard_03", "exit_extreme_00",
                                  "exit_extreme_01", "exit_extreme_02", "exit_extreme_03", "ml20m_per_movie_tall", "ml20m_per_user_tall"])
def test_block_kernel_matches_reference_fixture(device_solver, name):
    # lds limit 0 sends every entity through the team kernel, one workgroup per entity, L-BFGS vectors in HBM
    _solve_and_compare(device_solver, name, lds_limit=0)


@pytest.mark.parametrize("name", ["ref_fixture_l2_0.1", "ragged", "ml_per_user", "warm_stage2", "tiny_entities_regbias",
                                  "c2_no_intercept", "ragged_variance_simple", "c2_m3", "exit_factr_1e-4_m3_weights", "exit_extreme_00",
                                  "exit_extreme_02", "ml20m_per_movie_tall", "ml20m_per_user_tall"])
def test_device_wide_kernel_matches_reference_fixture(device_solver, name):
    # giant threshold 1 sends every entity, one after another, through the persistent device-wide kernel
    _solve_and_compare(device_solver, name, giant_nnz=1)


@pytest.mark.parametrize("name", ["ref_fixture_l2_0.1", "ragged", "ml_per_user", "warm_stage2", "c2_no_intercept",
                                  "ragged_variance_simple", "exit_factr_1e-7", "exit_hard_02", "exit_extreme_01", "exit_extreme_03",
                                  "ml20m_per_movie_tall", "ml20m_per_user_tall"])
def test_team_tiers_match_reference_fixture(device_solver, name):
    # every entity through the persistent kernel split into teams of CUs (2, 8 or 32 CUs by non-zeros)
    _solve_and_compare(device_solver, name, giant_nnz=0, team_nnz=1)


@pytest.mark.parametrize("dim,count_path", [(20, True), (24, False), (1000, True), (2048, True), (2049, True), (70000, True)])
def test_pack_of_large_entities_counting_and_sort_paths(device_solver, monkeypatch, dim, count_path):
    """csrc/re_pack_big.hip: entities above 1 024 non-zeros. Column indices below 2^11 take the counting path (chunk histograms, no
    sort), larger feature spaces the device-wide radix sort; GDMIX_PACK_BIG_COUNT=0 forces the sort path on a small feature space.
    Every array of the packed batch equals the oracle's, whichever path ran: entities of one chunk and of many (chunks of 1 024
    entries), a column that occurs once, empty samples, an entity with samples and no non-zero at all, small entities in between."""
    if count_path:
        monkeypatch.delenv("GDMIX_PACK_BIG_COUNT", raising=False)
    else:
        monkeypatch.setenv("GDMIX_PACK_BIG_COUNT", "0")
    from gdmix_amd.batch import RawBatch
    rng = np.random.default_rng(dim)
    ent_n = np.array([1500, 3, 9000, 40, 1100, 30000, 2, 1300, 5000, 150, 100, 300], np.int64)   # 1 100 samples of no entry: large by its samples
    ent_k = [(1, 5), (1, 4), (2, 4), (0, 9), (0, 1), (1, 3), (5, 6), (0, 3), (3, 4), (2, 5), (5, 9), (0, 2)]   # non-zeros per sample, [lo, hi)
    row_nnz = np.concatenate([rng.integers(lo, hi, n) for n, (lo, hi) in zip(ent_n, ent_k)])
    rp = np.concatenate([[0], np.cumsum(row_nnz)]).astype(np.int64)
    Z = int(rp[-1])
    # Zipf-like columns, plus the largest index exactly once (it sets the width of the column field)
    cols = np.minimum((float(dim) ** rng.random(Z)).astype(np.int64) - 1, dim - 1)
    cols[rng.integers(0, Z)] = dim - 1
    N = int(ent_n.sum())
    b = RawBatch(ent_row_ptr=np.concatenate([[0], np.cumsum(ent_n)]), row_nnz_ptr=rp, col_global=cols,
                 val=rng.standard_normal(Z).astype(np.float32), y=(rng.random(N) < 0.5).astype(np.float32),
                 offset=np.zeros(N, np.float32))
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    for _ in range(2):   # (the context keeps its temporaries between calls)
        packed = device_solver.pack(b)
        _check_pack(packed, pk, b.val)
        assert packed.max_p == int(np.diff(pk["ent_feat_ptr"]).max()) + 1
        assert packed.max_n == int(ent_n.max()) and packed.max_nnz == int(np.diff(rp[np.concatenate([[0], np.cumsum(ent_n)])]).max())


@pytest.mark.parametrize("dim", [3, 64, 1024, 2047, 2048, 2049, 70000])
@pytest.mark.parametrize("bitmap", [True, False])
def test_pack_of_small_entities_bitmap_and_rank_sort_paths(device_solver, monkeypatch, dim, bitmap):
    """csrc/re_pack.hip: entities of at most 128 non-zeros. All columns below 2 048: the bitmap path (local id = set bits below the
    column, CSC slot = the column's start + the rank among its own entries) — C2's and MovieLens' bags; a wider feature space, or
    GDMIX_PACK_BITMAP=0: the rank sort. Every array of the packed batch equals the oracle's, whichever path ran: one tile and two
    (more than 64 non-zeros), more than 64 distinct columns, a handful of columns each occurring many times (the ranking among equal
    columns, carried over the tile border), empty samples, entities without any non-zero, a single entry, exactly 128."""
    if bitmap:
        monkeypatch.delenv("GDMIX_PACK_BITMAP", raising=False)
    else:
        monkeypatch.setenv("GDMIX_PACK_BITMAP", "0")
    from gdmix_amd.batch import RawBatch
    rng = np.random.default_rng(1000 + dim)
    E = 600
    ent_n = rng.integers(1, 33, E).astype(np.int64)
    ent_n[:6] = [1, 128, 1, 40, 64, 16]
    row_nnz = []
    for e in range(E):
        budget, hi = 128, int(rng.integers(1, 9))
        k = np.minimum(rng.integers(0, hi + 1, ent_n[e]), 8)
        while k.sum() > budget:
            k[rng.integers(0, k.size)] = 0
        row_nnz.append(k)
    row_nnz[0] = np.array([1]); row_nnz[1] = np.ones(128, np.int64); row_nnz[2] = np.array([0]); row_nnz[3] = np.full(40, 3); row_nnz[4] = np.full(64, 2)
    row_nnz[5] = np.full(16, 8)
    row_nnz = np.concatenate(row_nnz).astype(np.int64)
    rp = np.concatenate([[0], np.cumsum(row_nnz)]).astype(np.int64)
    Z, N = int(rp[-1]), int(ent_n.sum())
    # a third of the entities draw from a handful of columns (many equal columns), the rest Zipf-like over the whole space
    ent_of_row = np.repeat(np.arange(E), ent_n)
    ent_of_nz = np.repeat(ent_of_row, row_nnz)
    few = (ent_of_nz % 3) == 0
    cols = np.minimum((float(dim) ** rng.random(Z)).astype(np.int64) - 1, dim - 1)
    cols[few] = rng.integers(0, min(dim, 5), int(few.sum()))
    cols[rng.integers(0, Z)] = dim - 1
    b = RawBatch(ent_row_ptr=np.concatenate([[0], np.cumsum(ent_n)]), row_nnz_ptr=rp, col_global=cols,
                 val=rng.standard_normal(Z).astype(np.float32), y=(rng.random(N) < 0.5).astype(np.float32),
                 offset=np.zeros(N, np.float32))
    nnz_e = np.diff(rp[b.ent_row_ptr])
    assert nnz_e.max() == 128 and (nnz_e > 64).sum() > 20 and (nnz_e == 0).any()
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    if dim >= 1024:
        assert np.diff(pk["ent_feat_ptr"]).max() > 64       # ids of the second half exist
    packed = device_solver.pack(b)
    _check_pack(packed, pk, b.val)
    assert packed.max_p == int(np.diff(pk["ent_feat_ptr"]).max()) + 1


def test_side_stream_does_not_change_results(device_solver, monkeypatch):
    """gdmix_re_solve launches the classes that cannot fill the device on a second stream of the context, beside the large ones
    (csrc/re_api.hip: class_is_small). A batch with a dozen classes, most of them small: a context without the side stream
    (GDMIX_RE_SIDE_STREAM=0 at creation) gives the same bits, SIMPLE variances included."""
    from gdmix_amd.batch import concat
    from gdmix_amd.solver import REDeviceSolver
    b = concat([synthetic.make_batch(3000, 16, 4, 1024, seed=61), synthetic.make_batch(200, 24, 8, 4096, seed=62, size_dist="zipf"),
                synthetic.make_movielens_20m("per_user", seed=63, entities=400), synthetic.make_batch(3, 900, 8, 4096, seed=64, size_dist="const")])
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12, variance_mode=1)
    packed = device_solver.pack(b)
    a = device_solver.solve(packed, SolverOptions(**kw)).to_host()
    assert sum(1 for _, c in device_solver.class_counts(packed) if c > 0) >= 8
    monkeypatch.setenv("GDMIX_RE_SIDE_STREAM", "0")
    plain = REDeviceSolver(0)
    try:
        r = plain.solve(plain.pack(b), SolverOptions(**kw)).to_host()
    finally:
        plain.close()
    for k in ("theta", "variance", "fval", "gnorm", "nit", "nfev", "status"):
        assert np.array_equal(a[k], r[k]), k


def test_unique_ids_compacted_next_to_the_solve_are_the_same(device_solver):
    """ABI 11, gdmix_re_set_defer_unique (csrc/re_pack.hip: the end of pack_impl): the compaction of the per-entity unique feature ids
    (job_consumers.py:243, np.unique) runs on a side stream next to the solve that follows the pack. The array is bit-exact against the
    oracle whether it is read right after the pack (the accessor waits), after the solve, or from a context that keeps the kernel inside
    the pack; batches packed and dropped without a solve, workspace reused at once, leave nothing behind; solves are the same bits."""
    import torch
    from gdmix_amd.solver import REDeviceSolver
    b = synthetic.make_batch(200_000, 16, 4, 1024, seed=71)
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
    rd = device_solver.upload(b)
    # (a) read right after the pack
    packed = device_solver.pack(rd)
    assert np.array_equal(packed.unique_global().cpu().numpy(), pk["unique_global"])
    # (b) read after the solve only, the pack -> solve pair as the product path issues it, a few times over with churn in between:
    # batches packed and dropped unsolved, their workspace handed straight to the next allocation
    want = None
    for rep in range(4):
        for _ in range(3):
            junk = device_solver.pack(rd)
            del junk
            torch.full((b.Z + 1,), -1, dtype=torch.int64, device="cuda")   # lands where a dropped workspace was
        packed = device_solver.pack(rd)
        res = device_solver.solve(packed, SolverOptions(**kw))
        got_u = packed.unique_global().cpu().numpy()
        assert np.array_equal(got_u, pk["unique_global"]), rep
        got = res.to_host()
        if want is None:
            want = got
        for k in ("theta", "fval", "nit", "nfev", "status"):
            assert np.array_equal(got[k], want[k]), (rep, k)
    # (c) a context that keeps the compaction inside the pack
    plain = REDeviceSolver(0)
    try:
        plain.set_defer_unique(False)
        pp = plain.pack(rd)
        r = plain.solve(pp, SolverOptions(**kw)).to_host()
        assert np.array_equal(pp.unique_global().cpu().numpy(), pk["unique_global"])
    finally:
        plain.close()
    for k in ("theta", "fval", "nit", "nfev", "status"):
        assert np.array_equal(want[k], r[k]), k


def test_large_classes_side_by_side_do_not_change_results(device_solver):
    """Round 4: the size classes that fill the device are dealt over the caller's stream and the context's side streams
    (gdmix_re_set_spread, default 4) instead of running one after another. 120 k C2 entities (three large group classes) with
    MovieLens-20M users (large tall classes) and a few small classes: two, three and four queues give the bits of the one-queue
    schedule, SIMPLE variances included, and the per-class times come back for every class that ran."""
    from gdmix_amd.batch import concat
    b = concat([synthetic.make_survey_batch(120_000, 16, 4, 1024, seed=65), synthetic.make_movielens_20m("per_user", seed=66, entities=9000),
                synthetic.make_batch(150, 24, 8, 4096, seed=67, size_dist="zipf")])
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12, variance_mode=1)
    packed = device_solver.pack(b)
    device_solver.set_timing(True)
    try:
        device_solver.set_spread(0)
        one = device_solver.solve(packed, SolverOptions(**kw)).to_host()
        ms_one = np.array(device_solver.last_solve_ms())
        counts = np.array([c for _, c in device_solver.class_counts(packed)])
        for q in (2, 3, 4):
            device_solver.set_spread(q)
            r = device_solver.solve(packed, SolverOptions(**kw)).to_host()
            ms = np.array(device_solver.last_solve_ms())
            for k in ("theta", "variance", "fval", "gnorm", "nit", "nfev", "status"):
                assert np.array_equal(one[k], r[k]), (q, k)
            assert np.array_equal(ms > 0, ms_one > 0)
    finally:
        device_solver.set_spread(4)
        device_solver.set_timing(False)
    assert (ms_one > 0).sum() >= 6 and counts[ms_one > 0].min() > 0


def test_team_lds_vectors_do_not_change_results(device_solver, monkeypatch):
    """The one-workgroup team kernel keeps x, g, d (and x_old, g_old where they fit) and the residuals of an entity in the LDS
    its workgroup leaves free (csrc/re_solve.hip: team_vec_level) instead of the global scratch slot. Same arithmetic in the
    same order: with the arena cut to 0 KB (everything in the slot), to 24 KB (x, g, d of the small ones only) and at its full
    size the bits are the same — entities of p from a few dozen to ~9 000, SIMPLE variances, a warm start."""
    from gdmix_amd.batch import concat
    b = concat([synthetic.make_batch(60, 40, 8, 65536, seed=71, size_dist="zipf"), synthetic.make_batch(6, 600, 8, 65536, seed=72, size_dist="const"),
                synthetic.make_batch(3, 1200, 8, 65536, seed=73, size_dist="const"), synthetic.make_batch(40, 12, 4, 256, seed=74)])
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12, variance_mode=1)
    device_solver.set_wave_lds_limit(0)       # every entity through the one-workgroup team kernel
    try:
        packed = device_solver.pack(b)
        p = np.diff(packed.coef_ptr_host())
        assert p.min() < 200 and 3200 < np.sort(p)[-9] < 5200 and p.max() > 8000      # placements 5, 3 and 1 occur at the full arena (0: arena cut)
        theta0 = np.random.default_rng(5).normal(0, 0.05, packed.P)
        runs = {}
        for kb in ("full", "24", "0"):
            if kb == "full":
                monkeypatch.delenv("GDMIX_TEAM_ARENA_KB", raising=False)
            else:
                monkeypatch.setenv("GDMIX_TEAM_ARENA_KB", kb)
            runs[kb] = (device_solver.solve(packed, SolverOptions(**kw)).to_host(), device_solver.solve(packed, SolverOptions(**kw), theta0=theta0).to_host())
    finally:
        device_solver.set_wave_lds_limit(65536)
    for kb in ("24", "0"):
        for cold_warm in (0, 1):
            for k in ("theta", "theta_thr", "variance", "fval", "gnorm", "nit", "nfev", "status"):
                assert np.array_equal(runs["full"][cold_warm][k], runs[kb][cold_warm][k]), (kb, cold_warm, k)
    assert runs["full"][0]["nit"].max() > 5 and np.all(runs["full"][0]["status"] <= 2)


def test_tall_kernels_give_the_same_bits_under_concurrent_load(device_solver):
    """The tall kernels add a sample's products into per-wavefront column accumulators with ds_add_f64, sixteen (or 32) sets per
    column indexed by lane, so lanes l, l + 16, l + 32, l + 48 of one instruction add to the same address: their bit reproducibility
    rests on the LDS applying the lanes of an instruction in lane order and a wavefront's instructions in issue order — observed
    on gfx950, not an architectural promise (csrc/re_solve_tall.hip documents the assumption). This test is the tripwire:
    MovieLens-20M-shaped entities (all three tall variants) solved twelve times, alone and while two other contexts on their own
    streams keep the device busy with C2-shaped solves — every run must give the same bits."""
    import threading
    from gdmix_amd.solver import REDeviceSolver
    import torch
    b = concat_ml = None
    from gdmix_amd.batch import concat
    b = concat([synthetic.make_movielens_20m("per_user", seed=81, entities=900), synthetic.make_movielens_20m("per_movie", seed=82, entities=2500)])
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
    packed = device_solver.pack(b)
    first = device_solver.solve(packed, SolverOptions(**kw)).to_host()
    counts = dict(device_solver.class_counts(packed))
    assert all(counts[k] > 0 for k in ("re_solve_tall_kernel<8> p<=64", "re_solve_tall_kernel<1> p<=64", "re_solve_tall_kernel<1> lean p<=64"))
    stop = threading.Event()

    def load(seed):
        s2 = REDeviceSolver(0)
        st = torch.cuda.Stream()
        noise = synthetic.make_batch(60_000, 16, 4, 1024, seed=seed)
        with torch.cuda.stream(st):
            pk = s2.pack(noise)
            while not stop.is_set():
                s2.solve(pk, SolverOptions(**kw))
                st.synchronize()
        s2.close()
    runs = []
    for phase in ("alone", "loaded"):
        threads = [threading.Thread(target=load, args=(90 + i,)) for i in range(2)] if phase == "loaded" else []
        for t in threads:
            t.start()
        try:
            for _ in range(6):
                runs.append(device_solver.solve(packed, SolverOptions(**kw)).to_host())
        finally:
            stop.set()
            for t in threads:
                t.join()
    for r in runs:
        for k in ("theta", "fval", "gnorm", "nit", "nfev", "status"):
            assert np.array_equal(first[k], r[k]), k


def test_team_tiers_give_the_same_bits_run_after_run_under_load(device_solver):
    """The multi-workgroup team kernels hand x, the residuals and the partial sums from workgroup to workgroup through HBM/L2 at
    every barrier; since round 4 a team whose workgroups all sit on one XCD skips the L2 write-back of the release (measured
    placement, csrc/re_solve_team.hpp: team_placement). A stale read would not crash, it would change a sum: the tripwire is that a
    batch solved twenty times — alone, and while two other contexts keep the device busy (uneven load, warm L1s) — gives the same
    bits every time, with the fast barrier and with the full one (GDMIX_RE_XCD_BARRIER=0) agreeing bit for bit as well."""
    import threading
    import torch
    from gdmix_amd.solver import REDeviceSolver
    b = synthetic.make_batch(1500, 32, 8, 65536, seed=91, size_dist="zipf")
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
    device_solver.set_team_nnz(64)           # [64, 512) -> up to 128 teams, [512, 8192) -> 32 teams, above -> 8 teams
    stop = threading.Event()

    def load(seed):
        s2 = REDeviceSolver(0)
        st = torch.cuda.Stream()
        noise = synthetic.make_batch(40_000, 16, 4, 1024, seed=seed)
        with torch.cuda.stream(st):
            pk = s2.pack(noise)
            while not stop.is_set():
                s2.solve(pk, SolverOptions(**kw))
                st.synchronize()
        s2.close()
    try:
        packed = device_solver.pack(b)
        first = device_solver.solve(packed, SolverOptions(**kw)).to_host()
        counts = dict(device_solver.class_counts(packed))
        assert counts["re_solve_team_kernel 128 teams"] > 50 and counts["re_solve_team_kernel 32 teams"] > 5, counts
        assert np.all(first["status"] <= 2)
        runs = []
        for phase in ("alone", "loaded"):
            threads = [threading.Thread(target=load, args=(95 + i,)) for i in range(2)] if phase == "loaded" else []
            for t in threads:
                t.start()
            try:
                for _ in range(10):
                    runs.append(device_solver.solve(packed, SolverOptions(**kw)).to_host())
            finally:
                stop.set()
                for t in threads:
                    t.join()
        os.environ["GDMIX_RE_XCD_BARRIER"] = "0"
        try:
            runs.append(device_solver.solve(packed, SolverOptions(**kw)).to_host())
        finally:
            os.environ.pop("GDMIX_RE_XCD_BARRIER", None)
    finally:
        device_solver.set_team_nnz(16384)
    for r in runs:
        for k in ("theta", "fval", "gnorm", "nit", "nfev", "status"):
            assert np.array_equal(first[k], r[k]), k


def test_contexts_with_persistent_grids_at_the_same_time(device_solver):
    """Round 5: three contexts of one process that all launch multi-workgroup team tiers (persistent grids that meet at barriers,
    one CU per workgroup) used to be able to split the device between two grids, each waiting for workgroups the other one's
    held — until the barrier's watchdog gave up and marked entities ABORTED (2 of 12.5 M on the full C5 share, with 5 s lost).
    The grids of a device are now chained whatever context they come from (ScopedGridGate): three threads, each with its own context
    and stream, solve a batch routed through the 128- and 32-team tiers ten times at once; every run ends with fmin_l_bfgs_b's own
    outcomes and the bits of the run done alone."""
    import threading
    import torch
    from gdmix_amd.solver import REDeviceSolver
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
    results, errors = {}, []

    def worker(i):
        try:
            s2 = REDeviceSolver(0)
            s2.set_team_nnz(64)
            st = torch.cuda.Stream()
            b = synthetic.make_batch(600, 32, 8, 65536, seed=191 + i, size_dist="zipf")
            with torch.cuda.stream(st):
                pk = s2.pack(b)
                alone.wait()              # (every context packed; the first solves below start together)
                out = []
                for _ in range(10):
                    out.append(s2.solve(pk, SolverOptions(**kw)).to_host())
                    st.synchronize()
            results[i] = (b, out, dict(s2.class_counts(pk)))
            s2.close()
        except Exception as e:      # noqa: BLE001
            errors.append(e)
            alone.abort()
    alone = threading.Barrier(3)
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    try:
        device_solver.set_team_nnz(64)
        for i in range(3):
            b, outs, counts = results[i]
            assert counts["re_solve_team_kernel 128 teams"] > 20 and counts["re_solve_team_kernel 32 teams"] > 2, counts
            ref = device_solver.solve(device_solver.pack(b), SolverOptions(**kw)).to_host()
            assert np.all(ref["status"] <= 2)
            for r in outs:
                for k in ("theta", "fval", "nit", "nfev", "status"):
                    assert np.array_equal(ref[k], r[k]), (i, k)
    finally:
        device_solver.set_team_nnz(16384)


def test_results_are_bitwise_reproducible(device_solver):
    b = synthetic.make_batch(2000, 16, 4, 1024, seed=5)
    packed = device_solver.pack(b)
    o = SolverOptions(regularize_bias=False)
    r1 = device_solver.solve(packed, o).to_host()
    r2 = device_solver.solve(packed, o).to_host()
    for k in ("theta", "fval", "nit", "nfev", "status"):
        assert np.array_equal(r1[k], r2[k]), k


def test_c2_shape_against_oracle_at_scale(device_solver):
    """20k C2-shaped entities (the bench workload's shape), shipped MovieLens options."""
    b = synthetic.make_batch(20000, 16, 4, 1024, seed=synthetic.C2_SEED)
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
    packed = device_solver.pack(b)
    res = device_solver.solve(packed, SolverOptions(**kw)).to_host()
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    assert np.array_equal(packed.unique_global().cpu().numpy(), pk["unique_global"])
    ref = oracle.solve(pk, b.val, b.y, b.offset, None, oracle.make_opts(**kw))
    coef_ptr = packed.coef_ptr_host()
    wp = well_posed_mask(b, dict(l2=1.0, regularize_bias=False, has_intercept=True))
    err = per_entity_rel_err(res["theta"], ref["theta"], coef_ptr)
    assert err[wp].max() <= REL_TOL_DEVICE
    assert (res["nit"][wp] == ref["nit"][wp]).mean() == 1.0
    assert np.isin(res["status"], (0, 1, 2)).all()


def test_c2_at_full_size_properties(device_solver):
    """BASELINE.json's C2 at its full size (1 M entities, 16 M samples, 64 M non-zeros: the bench workload, SURVEY 8(d) generator),
    where the oracle is too slow to compare everything: properties that do not depend on the size.
    * every entity ends with one of fmin_l_bfgs_b's own outcomes, the returned gradient norm of a PGTOL stop is below pgtol;
    * entities are independent: the same entities in another order (other neighbours in a wavefront, other positions in their size
      class) give bit-identical coefficients, iteration counts and stops;
    * a solve started from the answer stays there: zero iterations for every entity that had stopped on the gradient test, the
      coefficients unchanged bit for bit;
    * a sample of 3 000 entities against the oracle, iteration for iteration."""
    E = 1_000_000
    b = synthetic.make_survey_batch(E, 16, 4, 1024, seed=synthetic.C2_SEED)
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
    o = SolverOptions(**kw)
    packed = device_solver.pack(b)
    res = device_solver.solve(packed, o).to_host()
    coef_ptr = packed.coef_ptr_host()
    assert np.isin(res["status"], (0, 1, 2)).all() and (res["status"] == 0).mean() > 0.99
    assert res["gnorm"][res["status"] == 0].max() <= 1e-5
    assert 8.0 < res["nit"].mean() < 9.0 and 9.0 < res["nfev"].mean() < 10.0        # what SURVEY measured on the reference: 8.4 / 9.4
    # another order
    perm = np.random.default_rng(1).permutation(E)
    bp = b.select(perm)
    pp = device_solver.pack(bp)
    rp = device_solver.solve(pp, o).to_host()
    cp = pp.coef_ptr_host()
    assert np.array_equal(np.diff(cp), np.diff(coef_ptr)[perm])
    for k in ("nit", "nfev", "status", "fval"):
        assert np.array_equal(rp[k], res[k][perm]), k
    from gdmix_amd.batch import _ranges
    take = _ranges(coef_ptr[perm], np.diff(coef_ptr)[perm])
    assert np.array_equal(rp["theta"], res["theta"][take])
    del pp, rp, bp
    # started from the answer
    again = device_solver.solve(packed, o, theta0=res["theta"]).to_host()
    stopped = res["status"] == 0
    assert (again["nit"][stopped] == 0).all() and (again["status"][stopped] == 0).all()
    m = np.zeros(coef_ptr[-1], bool)
    m[_ranges(coef_ptr[:-1][stopped], np.diff(coef_ptr)[stopped])] = True
    assert np.array_equal(again["theta"][m], res["theta"][m])
    # a sample against the oracle
    sample = np.sort(np.random.default_rng(2).choice(E, 3000, replace=False))
    sb = b.select(sample)
    pk = oracle.pack(sb.ent_row_ptr, sb.row_nnz_ptr, sb.col_global)
    ref = oracle.solve(pk, sb.val, sb.y, sb.offset, None, oracle.make_opts(**kw))
    wp = well_posed_mask(sb, dict(l2=1.0, regularize_bias=False, has_intercept=True))
    sub_theta = res["theta"][_ranges(coef_ptr[sample], np.diff(coef_ptr)[sample])]
    sub_ptr = np.concatenate([[0], np.cumsum(np.diff(coef_ptr)[sample])])
    err = per_entity_rel_err(sub_theta, ref["theta"], sub_ptr)
    assert err[wp].max() <= REL_TOL_DEVICE
    assert np.array_equal(res["nit"][sample][wp], ref["nit"][wp]) and np.array_equal(res["status"][sample][wp], ref["status"][wp])


def test_large_and_giant_entities_pack_and_solve(device_solver):
    """Entities beyond the wavefront pack/solve paths: a Zipf tail with one ~50k-nnz entity (workgroup pack
    kernel, workgroup-per-entity solve out of global scratch), pack bit-exact vs the oracle, theta vs oracle."""
    rng = np.random.default_rng(3)
    from gdmix_amd.batch import concat
    small = synthetic.make_batch(300, 24, 8, 4096, seed=21, size_dist="zipf")
    giant = synthetic.make_batch(2, 6000, 8, 4096, seed=22, size_dist="const")
    mid = synthetic.make_batch(6, 300, 8, 4096, seed=23, size_dist="const")
    b = concat([small, giant, mid])
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    packed = device_solver.pack(b)
    _check_pack(packed, pk, b.val)
    assert packed.max_nnz == int(b.ent_nnz().max()) and packed.max_n == int(b.ent_n().max())
    ref = oracle.solve(pk, b.val, b.y, b.offset, None, oracle.make_opts(**kw))
    coef_ptr = packed.coef_ptr_host()
    wp = well_posed_mask(b, dict(l2=1.0, regularize_bias=False, has_intercept=True))
    # the two 48k-nnz entities through each tier: workgroup kernel, 128 / 32 / 8 teams, device-wide kernel
    for giant_nnz, team_nnz, mask, cls in ((0, 0, 7, "re_solve_team_kernel workgroup"), (16777216, 16384, 7, "re_solve_team_kernel 128 teams"),
                                           (16777216, 4096, 7, "re_solve_team_kernel 32 teams"), (16777216, 256, 7, "re_solve_team_kernel 8 teams"),
                                           (40000, 0, 7, "re_solve_team_kernel device-wide")):
        device_solver.set_giant_nnz(giant_nnz)
        device_solver.set_team_nnz(team_nnz)
        device_solver.set_kernel_mask(mask)
        try:
            res = device_solver.solve(packed, SolverOptions(**kw)).to_host()
        finally:
            device_solver.set_giant_nnz(16777216)
            device_solver.set_team_nnz(16384)
            device_solver.set_kernel_mask(7)
        counts = dict(device_solver.class_counts(packed))
        assert counts[cls] >= 2
        err = per_entity_rel_err(res["theta"], ref["theta"], coef_ptr)
        assert err[wp].max() <= REL_TOL_DEVICE, err[wp].max()
        assert np.array_equal(res["nit"][wp], ref["nit"][wp])
        assert np.array_equal(res["status"][wp], ref["status"][wp])


def test_full_variance_of_entities_beyond_one_wavefront(device_solver):
    """variance_mode FULL where an entity has more than 2048 coefficients (csrc/re_variance_big.hip: Hessian, tiled Cholesky and
    inverse on the whole device, one entity at a time) next to small entities (one wavefront each): against the dense statement
    of binary_logistic_regression.py:181-187 in numpy — toarray() sums repeated columns of a row, inv() of the full Hessian."""
    from gdmix_amd.batch import concat
    small = synthetic.make_ragged_batch(40, seed=5, D=300)
    large = synthetic.make_batch(2, 1500, 8, 2600, seed=31, size_dist="const")
    tail = synthetic.make_batch(3, 50, 6, 504, seed=32, size_dist="const")
    b = concat([small, large, tail])
    for regularize_bias in (False, True):
        kw = dict(l2=0.7, regularize_bias=regularize_bias, has_intercept=True, m=10, max_iter=25, variance_mode=2)
        packed = device_solver.pack(b)
        assert packed.max_p > 2048
        res = device_solver.solve(packed, SolverOptions(**kw)).to_host()
        coef_ptr = packed.coef_ptr_host()
        pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
        checked_large = 0
        for e in range(b.E):
            r0, r1 = b.ent_row_ptr[e], b.ent_row_ptr[e + 1]
            f0 = pk["ent_feat_ptr"][e]
            d = int(pk["ent_feat_ptr"][e + 1] - f0)
            p = d + 1
            if p <= 2048 and e % 7:
                continue
            uniq = pk["unique_global"][f0:f0 + d]
            X = np.zeros((r1 - r0, p))
            X[:, 0] = 1.0
            for i in range(r0, r1):
                k0, k1 = b.row_nnz_ptr[i], b.row_nnz_ptr[i + 1]
                np.add.at(X[i - r0], 1 + np.searchsorted(uniq, b.col_global[k0:k1]), b.val[k0:k1].astype(np.float64))
            th = res["theta"][coef_ptr[e]:coef_ptr[e + 1]]
            rho = 1.0 / (1.0 + np.exp(-(X @ th + b.offset[r0:r1])))
            w = b.weight[r0:r1] if b.weight is not None else 1.0
            H = (X * (rho * (1 - rho) * w)[:, None]).T @ X + (0.7 + 1e-12) * np.eye(p)
            if not regularize_bias:
                H[0, 0] -= 0.7
            want = np.diag(np.linalg.inv(H))
            np.testing.assert_allclose(res["variance"][coef_ptr[e]:coef_ptr[e + 1]], want, rtol=1e-8)
            checked_large += p > 2048
        assert checked_large == 2


def test_score_matches_reference_inference(device_solver):
    import os
    from helpers import GOLDEN
    b, opts, exp, _ = load_fixture("warm_stage2")
    z = np.load(os.path.join(GOLDEN, "score_stage2.npz"))
    packed = device_solver.pack(b)
    has_model = np.zeros(b.E, np.uint8)
    has_model[:int(z["n_with_model"])] = 1
    logit, per = device_solver.score(packed, exp["theta_thr"], has_model)
    logit, per = logit.cpu().numpy(), per.cpu().numpy()
    np.testing.assert_allclose(logit, z["exp_score"].astype(np.float32), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(per, z["exp_per_coord"].astype(np.float32), rtol=1e-5, atol=1e-6)
    r0 = b.ent_row_ptr[int(z["n_with_model"])]
    assert np.array_equal(logit[r0:], b.offset[r0:])
    assert np.all(per[r0:] == 0)
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    lo, po = oracle.score(pk, b.val, b.offset, exp["theta_thr"], True, has_model)
    np.testing.assert_allclose(logit, lo, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("shape", ["ragged", "zipf", "single_giant", "tiny"])
@pytest.mark.parametrize("has_intercept", [True, False])
def test_score_any_entity_sizes_matches_oracle(device_solver, shape, has_intercept):
    """The scoring pass is one thread per sample over the whole batch: entities of one sample next to entities of a
    hundred thousand, empty rows, entities without a model — against the CPU restatement, to float rounding."""
    from gdmix_amd import synthetic
    if shape == "ragged":
        b = synthetic.make_ragged_batch(5000, seed=21)
    elif shape == "zipf":
        b = synthetic.make_batch(3000, 32, 8, 65536, seed=5, size_dist="zipf", with_uid=False)
    elif shape == "single_giant":
        b = synthetic.make_batch(1, 100001, 8, 4096, seed=6, size_dist="const")
    else:
        b = synthetic.make_batch(3, 1, 2, 16, seed=7, size_dist="const")
    packed = device_solver.pack(b, has_intercept=has_intercept)
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    rng = np.random.default_rng(8)
    theta = rng.standard_normal(int(packed.P))
    has_model = (rng.random(b.E) < 0.8).astype(np.uint8)
    for hm in (None, has_model):
        logit, per = device_solver.score(packed, theta, hm)
        lo, po = oracle.score(pk, b.val, b.offset, theta, has_intercept, hm)
        np.testing.assert_allclose(logit.cpu().numpy(), lo.astype(np.float32), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(per.cpu().numpy(), po.astype(np.float32), rtol=2e-5, atol=2e-5)


def test_partition_ids_bit_exact(device_solver):
    rng = np.random.default_rng(0)
    ids = np.concatenate([rng.integers(-2**62, 2**62, size=5000), np.arange(-50, 2000),
                          [0, -1, 2**63 - 1, -2**63, 100034, 943, 1682]]).astype(np.int64)
    for n in (1, 10, 1024, 2**31 - 1):
        got = device_solver.partition_ids(ids, n).cpu().numpy()
        want = np.array([oracle.java_partition_id(str(int(v)), n) for v in ids], np.int32)
        assert np.array_equal(got, want)


def test_empty_and_single_entity_batches(device_solver):
    from gdmix_amd.batch import RawBatch
    one = RawBatch(ent_row_ptr=[0, 1], row_nnz_ptr=[0, 1], col_global=[3], val=[2.0], y=[1], offset=[0.0])
    packed = device_solver.pack(one)
    res = device_solver.solve(packed, SolverOptions()).to_host()
    pk = oracle.pack(one.ent_row_ptr, one.row_nnz_ptr, one.col_global)
    ref = oracle.solve(pk, one.val, one.y, one.offset, None, oracle.make_opts())
    np.testing.assert_allclose(res["theta"], ref["theta"], rtol=1e-9)
    assert res["nit"][0] == ref["nit"][0]


@pytest.mark.parametrize("shape", ["ragged", "c2", "wide_rows", "tiny", "float_labels"])
def test_wire_form_widens_to_the_raw_batch(device_solver, shape):
    """gdmix_re_wire_batch (counts, int32 feature ids, byte labels) -> gdmix_re_widen gives bit for bit the raw arrays a
    direct upload would, for every count width; the pack that follows is the same pack."""
    import torch
    if shape == "ragged":
        b = synthetic.make_ragged_batch(3000, seed=31)          # empty samples, duplicates, weights
    elif shape == "c2":
        b = synthetic.make_batch(50000, 16, 4, 1024, seed=32)
    elif shape == "wide_rows":                                  # more than 255 (and 65535) non-zeros in a sample
        from gdmix_amd.batch import RawBatch
        rng = np.random.default_rng(33)
        k = np.array([3, 300, 0, 70000, 1, 0, 2], np.int64)
        rp = np.concatenate([[0], np.cumsum(k)])
        b = RawBatch(ent_row_ptr=[0, 2, 2, 5, 7], row_nnz_ptr=rp, col_global=rng.integers(0, 2**31 - 1, rp[-1]),
                     val=rng.standard_normal(rp[-1]), y=[0, 1, 1, 0, 1, 0, 0], offset=rng.standard_normal(7))
    elif shape == "tiny":
        from gdmix_amd.batch import RawBatch
        b = RawBatch(ent_row_ptr=[0, 1], row_nnz_ptr=[0, 1], col_global=[3], val=[2.0], y=[1], offset=[0.0])
    else:
        b = synthetic.make_batch(2000, 16, 4, 1024, seed=34)
        b.y = (b.y * 0.25 + 0.1).astype(np.float32)
        b.binary_labels = False
    w = b.to_wire()
    assert w["row_nnz_width"] == {"ragged": 1, "c2": 1, "wide_rows": 4, "tiny": 1, "float_labels": 1}[shape]
    assert w["col_width"] == (4 if shape == "wide_rows" else 2)
    rd = device_solver.widen(device_solver.upload_wire(w))
    for k in ("ent_row_ptr", "row_nnz_ptr", "col_global", "val", "y", "offset"):
        assert np.array_equal(rd[k].cpu().numpy(), getattr(b, k)), k
    assert (rd["weight"] is None) == (b.weight is None)
    if b.weight is not None:
        assert np.array_equal(rd["weight"].cpu().numpy(), b.weight)
    if shape in ("ragged", "c2"):
        pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
        _check_pack(device_solver.pack(rd), pk, b.val)
    # through page-locked staging blocks, as the pipelined hand-over does
    pinned = {k: torch.empty(max(1, 0 if w[k] is None else w[k].size), dtype=torch.from_numpy(w[k]).dtype).pin_memory()
              for k in device_solver.WIRE_ARRAYS if w[k] is not None}
    rd2 = device_solver.widen(device_solver.upload_wire(w, pinned=pinned))
    torch.cuda.synchronize()
    for k in ("ent_row_ptr", "row_nnz_ptr", "col_global", "y"):
        assert np.array_equal(rd2[k].cpu().numpy(), getattr(b, k)), k


def test_pinned_routing_makes_an_entity_independent_of_its_batch(device_solver):
    """by default the split between the tall kernels and the team threshold are chosen per batch, so the same entity can
    get another kernel — other last bits — in another batch. REDeviceSolver.pin_routing (what a model with rebalance_entities
    does) ties the kernel to the entity's own size: every fortieth of 24 000 MovieLens-20M users solved in a batch of 600 and in the batch of
    24 000 comes out bit for bit the same, the per-batch choice being demonstrably different for the two batches by default."""
    b = synthetic.make_movielens_20m("per_user", seed=83, entities=24000)     # too many users above 512 samples for a lowered split ...
    ents = np.arange(0, b.E, 40)                                               # ... 600 of them: a small batch, which gets one
    sub = b.select(ents)
    opts = SolverOptions(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)

    def both():
        out = []
        for batch in (b, sub):
            packed = device_solver.pack(batch)
            r = device_solver.solve(packed, opts).to_host()
            cp = packed.coef_ptr_host()
            cls = packed._view(packed.c.cls_tmp, packed.E, device_solver.torch.int32).cpu().numpy().copy()
            out.append((r, cp, cls))
        return out
    (rb, cpb, clsb), (rs, cps, clss) = both()
    differently_routed = int((clsb[ents] != clss).sum())
    try:
        device_solver.pin_routing()
        (rb, cpb, clsb), (rs, cps, clss) = both()
    finally:
        device_solver.set_tall_split_n(0)
        device_solver.set_tall_team_n(device_solver.TALL_TEAM_N_DEFAULT)
    assert differently_routed > 0            # (otherwise this shape no longer exercises the per-batch choice: pick another)
    assert np.array_equal(clsb[ents], clss)
    for i, e in enumerate(ents):
        assert np.array_equal(rb["theta"][cpb[e]:cpb[e + 1]], rs["theta"][cps[i]:cps[i + 1]]), e
    assert np.array_equal(rb["nit"][ents], rs["nit"]) and np.array_equal(rb["status"][ents], rs["status"])


def test_a_small_batch_lowers_the_tall_split_and_only_the_rounding_changes(device_solver):
    """Round 4: a batch whose eight-wavefront tall class stays small anyway (a share of a strongly scaled MovieLens job) gets a lower
    split between the one-wavefront and the eight-wavefront tall kernels — chosen on the device (class_base_kernel), the entities moved
    by re_order_kernel. 1 500 MovieLens-20M users: with the split fixed (an explicit gdmix_re_set_tall_split_n keeps it) only the few
    users above 4 096 samples take the eight-wavefront kernel; by default every user of at least the chosen split does, the counts add
    up, nobody is lost, and the two solutions agree to rounding (different summation order; same iteration counts but for the
    entities whose trajectory is rounding-sensitive) — both against the oracle's tolerance of the fixtures."""
    b = synthetic.make_movielens_20m("per_user", seed=83, entities=1500)
    n = b.ent_n()
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
    names = ("re_solve_tall_kernel<8> p<=64", "re_solve_tall_kernel<1> p<=64", "re_solve_tall_kernel<1> lean p<=64")
    packed = device_solver.pack(b)
    try:
        device_solver.set_tall_mid_n(-1)          # the per-batch mid class (off by default: it did not pay on these shares) with the split's
        adaptive = device_solver.solve(packed, SolverOptions(**kw)).to_host()
    finally:
        device_solver.set_tall_mid_n(0)
    ca = dict(device_solver.class_counts(packed))
    cls_a = packed._view(packed.c.cls_tmp, packed.E, device_solver.torch.int32).cpu().numpy().copy()
    try:
        device_solver.set_tall_split_n(4097)      # any explicit value switches the per-batch choice off
        fixed = device_solver.solve(packed, SolverOptions(**kw)).to_host()
        cf = dict(device_solver.class_counts(packed))
        cls_f = packed._view(packed.c.cls_tmp, packed.E, device_solver.torch.int32).cpu().numpy().copy()
    finally:
        device_solver.set_tall_split_n(0)
    assert cf[names[0]] == int((n >= 4097).sum()) and cf[names[0]] < 20
    moved = ca[names[0]] - cf[names[0]]
    assert moved > 20 and ca[names[0]] <= 384, (ca[names[0]], cf[names[0]])
    # round 6: of what stays below the split, the largest go to the MID class (four wavefronts, two workgroups per CU) — at most one
    # round of its launch, from the lowest of its thresholds that fits; a pinned split (the fixed run) pins that choice too
    mid_name = "re_solve_tall_kernel<4> p<=64"
    assert cf[mid_name] == 0 and 0 < ca[mid_name] <= 512
    assert ca[names[1]] + ca[mid_name] == cf[names[1]] - moved and ca[names[2]] == cf[names[2]]
    assert sum(ca.values()) == sum(cf.values()) == b.E
    # exactly the one-wavefront entities of at least the chosen split moved, and the split is one of the three candidates
    idx = {name: i for i, (name, _) in enumerate(device_solver.class_counts(packed))}
    went = (cls_a == idx[names[0]]) & (cls_f != idx[names[0]])
    assert int(went.sum()) == moved
    split = [s for s in (512, 1024, 2048) if np.array_equal(went, (cls_f == idx[names[1]]) & (n >= s))]
    assert split, int(n[went].min())
    went_mid = cls_a == idx[mid_name]
    steps = [t for t in (256, 384, 512, 768, 1024, 1536) if t < split[0]]
    fits = [t for t in steps if np.array_equal(went_mid, (cls_f == idx[names[1]]) & (n >= t) & (n < split[0]))]
    assert fits, (int(n[went_mid].min()), int(went_mid.sum()), split)
    # ... the lowest that fits: the next lower threshold would have made the class larger than one round
    lower = [t for t in steps if t < fits[0]]
    assert all(int(((cls_f == idx[names[1]]) & (n >= t) & (n < split[0])).sum()) > 512 for t in lower)
    assert np.array_equal(adaptive["status"], fixed["status"])
    same_nit = adaptive["nit"] == fixed["nit"]
    assert same_nit.mean() > 0.97
    cp = packed.coef_ptr_host()
    worst = 0.0
    for e in np.flatnonzero(same_nit):
        a, f = adaptive["theta"][cp[e]:cp[e + 1]], fixed["theta"][cp[e]:cp[e + 1]]
        worst = max(worst, float(np.max(np.abs(a - f)) / max(np.max(np.abs(f)), 1e-300)))
    assert worst <= 1e-7, worst
    np.testing.assert_allclose(adaptive["fval"], fixed["fval"], rtol=1e-9, atol=1e-12)


def test_the_tallest_entities_of_a_batch_get_a_team_of_workgroups(device_solver, monkeypatch):
    """Round 4: the team class (four workgroups, four CUs of one XCD, on one entity) takes the eight-wavefront tall entities above
    the lowest of 8 192 / 16 384 / 32 768 samples that keeps it within a quarter of the CUs' worth of entities (64 on an MI355X) — chosen on the device
    (class_base_kernel), the entities moved by re_order_kernel. 3 300 MovieLens-20M movies (a share of eight): the counts add up, exactly
    the entities above the chosen threshold moved, and the solution agrees with the one-workgroup kernel's to rounding; switched off
    (team_n 0) nobody moves; the same bits run after run, and with the full release instead of the same-XCD signals."""
    b = synthetic.make_movielens_20m("per_movie", seed=84, entities=3300)
    n = b.ent_n()
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
    team, tall8 = "re_solve_tall_team_kernel<8> x4 p<=64", "re_solve_tall_kernel<8> p<=64"
    packed = device_solver.pack(b)
    try:
        device_solver.set_tall_split_n(4097)     # an explicit split: the per-batch choice between the <1> and <8> kernels stays out of the comparison
        with_team = device_solver.solve(packed, SolverOptions(**kw, variance_mode=1)).to_host()
        ct = dict(device_solver.class_counts(packed))
        idx = {name: i for i, (name, _) in enumerate(device_solver.class_counts(packed))}
        cls_t = packed._view(packed.c.cls_tmp, packed.E, device_solver.torch.int32).cpu().numpy().copy()
        again = device_solver.solve(packed, SolverOptions(**kw, variance_mode=1)).to_host()
        monkeypatch.setenv("GDMIX_RE_XCD_BARRIER", "0")
        full_release = device_solver.solve(packed, SolverOptions(**kw, variance_mode=1)).to_host()
        monkeypatch.delenv("GDMIX_RE_XCD_BARRIER")
        device_solver.set_tall_team_n(0)
        without = device_solver.solve(packed, SolverOptions(**kw, variance_mode=1)).to_host()
        cw = dict(device_solver.class_counts(packed))
    finally:
        device_solver.set_tall_team_n(device_solver.TALL_TEAM_N_DEFAULT)
        device_solver.set_tall_split_n(0)
    assert cw[team] == 0 and sum(cw.values()) == sum(ct.values()) == b.E
    limit = 64
    assert 0 < ct[team] <= limit and ct[team] + ct[tall8] == cw[tall8], (ct[team], ct[tall8], cw[tall8])
    went = cls_t == idx[team]
    assert [t for t in (8192, 16384, 32768) if np.array_equal(went, n >= t)], (int(n[went].min()), int(went.sum()))
    # the lowest threshold that fits: the next lower one would not have
    t = int(n[went].min())
    lower = [x for x in (8192, 16384) if x * 2 <= t]
    assert all(int((n >= x).sum()) > limit for x in lower)
    for k in ("theta", "variance", "fval", "gnorm", "nit", "nfev", "status"):
        assert np.array_equal(with_team[k], again[k]), k
        assert np.array_equal(with_team[k], full_release[k]), k
    assert np.all(with_team["status"] <= 2) and np.array_equal(with_team["status"], without["status"])
    same_nit = with_team["nit"] == without["nit"]
    assert same_nit[went].mean() >= 0.9 and np.all(same_nit[~went])
    cp = packed.coef_ptr_host()
    worst = 0.0
    for e in np.flatnonzero(same_nit):
        a, f = with_team["theta"][cp[e]:cp[e + 1]], without["theta"][cp[e]:cp[e + 1]]
        worst = max(worst, float(np.max(np.abs(a - f)) / max(np.max(np.abs(f)), 1e-300)))
    assert worst <= 1e-7, worst
    np.testing.assert_allclose(with_team["fval"], without["fval"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(with_team["variance"][np.repeat(same_nit, np.diff(cp))], without["variance"][np.repeat(same_nit, np.diff(cp))], rtol=1e-6)
    # everybody outside the class: bit for bit what the run without the class gave
    keep = np.repeat(~went, np.diff(cp))
    assert np.array_equal(with_team["theta"][keep], without["theta"][keep])
