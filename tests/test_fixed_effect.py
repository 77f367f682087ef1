"""Fixed-effect objective (gdmix_re_opts.sum_loss / .linear): the oracle against fixtures produced by the reference's own
numpy + scipy ground truth (tests/golden/generate_fe_fixtures.py), and the device path against both."""
import glob
import os

import numpy as np
import pytest

from gdmix_amd import fixed_effect as fe
from gdmix_amd.solver import SolverOptions
from oracle import oracle
import fuzz_fe_case

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(HERE, "golden", "fe_*.npz")))
REL_TOL = 1e-9          # runs that stop on the projected-gradient test: same trajectory, 1e-16 .. 1e-12 observed
REL_TOL_FACTR = 1e-5    # the north-star bar. A run that stops on (f_old - f) <= 1e-12 * |f| (status 1) stops at rounding
                        # noise level: the iteration at which that fires, hence theta within the convergence radius,
                        # depends on summation order (linear_wide: 77 iterations, 2.5e-6 between scipy and the oracle)


def tol(status):
    return REL_TOL_FACTR if status == 1 else REL_TOL


def load(name):
    z = np.load(os.path.join(HERE, "golden", f"fe_{name}.npz"))
    return {k: z[k] for k in z.files}


def rel_err(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def oracle_fit(c):
    ic = bool(c["has_intercept"])
    D = int(c["num_features"])
    batch, dummy = fe.shard_as_batch(c["row_nnz_ptr"], c["col_global"], c["val"], c["y"], c["offset"], None, ic,
                                     binary_labels=not c["linear"])
    pk = oracle.pack(batch.ent_row_ptr, batch.row_nnz_ptr, batch.col_global)
    uniq = pk["unique_global"]
    t0 = fe.to_local(c["theta0"], uniq, D, ic, dummy) if c["theta0"].size else None
    o = oracle.make_opts(l2=float(c["l2"]), regularize_bias=ic, has_intercept=ic, m=10, max_iter=int(c["max_iter"]), ftol=1e-12,
                         threshold=0.0, sum_loss=True, linear=bool(c["linear"]))
    res = oracle.solve(pk, batch.val, batch.y, batch.offset, None, o, theta0=t0)
    return fe.to_global(res["theta"], uniq, D, ic, dummy), res


def test_fixtures_exist():
    assert len(NAMES) >= 10


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_the_reference_ground_truth(name):
    c = load(name)
    theta, res = oracle_fit(c)
    assert res["status"][0] in (0, 1, 2)
    assert rel_err(theta, c["theta"]) <= tol(res["status"][0]), rel_err(theta, c["theta"])
    if int(c["max_iter"]) == 100 and not c["theta0"].size and float(c["l2"]) == 1.0:
        # the reference's own expected coefficients for this seed (solved on float64 features, stored as float32)
        assert rel_err(theta.astype(np.float32), c["ref_expected_f32"]) <= 2e-3


def test_layout_mapping_round_trip():
    uniq = np.array([1, 4, 7])
    g = np.array([0.0, 2.0, 0.0, 0.0, 3.0, 0.0, 0.0, 4.0, 9.0])      # 8 features + intercept last
    loc = fe.to_local(g, uniq, 8, True, False)
    assert loc.tolist() == [9.0, 2.0, 3.0, 4.0]
    assert fe.to_global(loc, uniq, 8, True, False).tolist() == g.tolist()
    assert fe.to_global(np.array([2.0, 3.0, 4.0]), uniq, 8, False, False).tolist() == g[:8].tolist()
    b, dummy = fe.shard_as_batch([0, 0, 0], [], [], [1, 0], None, None, True)
    assert dummy and b.Z == 2 and not b.val.any()
    with pytest.raises(ValueError):
        fe.shard_as_batch([0, 0, 0], [], [], [1, 0], None, None, False)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_matches_reference_ground_truth_and_oracle(device_solver, name):
    c = load(name)
    ic = bool(c["has_intercept"])
    s = fe.FixedEffectDeviceSolver(solver=device_solver)
    theta, info = s.fit(c["row_nnz_ptr"], c["col_global"], c["val"], c["y"], int(c["num_features"]), offset=c["offset"],
                        has_intercept=ic, l2=float(c["l2"]), regularize_bias=True,
                        model_type=fe.LINEAR_REGRESSION if c["linear"] else fe.LOGISTIC_REGRESSION,
                        theta0=c["theta0"] if c["theta0"].size else None, max_iter=int(c["max_iter"]))
    assert info["status"] in (0, 1, 2)
    assert rel_err(theta, c["theta"]) <= tol(info["status"]), rel_err(theta, c["theta"])
    th_o, res = oracle_fit(c)
    assert rel_err(theta, th_o) <= tol(info["status"])
    assert info["status"] == res["status"][0]
    if info["status"] != 1:
        assert info["nit"] == res["nit"][0] and info["nfev"] == res["nfev"][0]


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_stepping_kernels_match_reference_ground_truth_and_oracle(device_solver, name):
    """include/gdmix_fe.h: streaming passes + replicated L-BFGS step (single worker: no all-reduce)."""
    c = load(name)
    ic = bool(c["has_intercept"])
    s = fe.FixedEffectDeviceSolver(solver=device_solver)
    theta, info = s.fit_stepping(c["row_nnz_ptr"], c["col_global"], c["val"], c["y"], int(c["num_features"]), offset=c["offset"],
                                 has_intercept=ic, l2=float(c["l2"]), regularize_bias=True,
                                 model_type=fe.LINEAR_REGRESSION if c["linear"] else fe.LOGISTIC_REGRESSION,
                                 theta0=c["theta0"] if c["theta0"].size else None, max_iter=int(c["max_iter"]))
    assert info["status"] in (0, 1, 2)
    assert rel_err(theta, c["theta"]) <= tol(info["status"]), rel_err(theta, c["theta"])
    th_o, res = oracle_fit(c)
    assert info["status"] == res["status"][0]
    if info["status"] != 1:
        assert info["nit"] == res["nit"][0] and info["nfev"] == res["nfev"][0]


@pytest.mark.gpu
def test_stepping_kernels_on_ragged_rows_long_columns_and_weights(device_solver):
    """Row lengths from 0 to more than a unit of the passes, a handful of very long columns, empty rows at both
    ends, weights, unregularised intercept: device-wide team kernel, stepping kernels and oracle agree."""
    rng = np.random.default_rng(5)
    n, D = 3000, 700
    k = rng.integers(0, 40, n)
    k[rng.integers(0, n, 5)] = rng.integers(5000, 12000, 5)      # rows longer than one 4096-entry block
    k[:3] = 0
    k[-2:] = 0
    rp = np.concatenate([[0], np.cumsum(k)]).astype(np.int64)
    cols = rng.integers(0, D, rp[-1])
    cols[rng.random(rp[-1]) < 0.5] = rng.integers(0, 3, int((rng.random(rp[-1]) < 0.5).sum()) or 1)[0]   # one dominant column
    vals = (rng.standard_normal(rp[-1]) * 0.1).astype(np.float32)
    y = (rng.random(n) < 0.4).astype(np.float32)
    off = (0.2 * rng.standard_normal(n)).astype(np.float32)
    wt = (0.5 + rng.random(n)).astype(np.float32)
    s = fe.FixedEffectDeviceSolver(solver=device_solver)
    kw = dict(offset=off, weight=wt, l2=2.0, regularize_bias=False, max_iter=60)
    th_team, info_team = s.fit(rp, cols, vals, y, D, **kw)
    th_step, info_step = s.fit_stepping(rp, cols, vals, y, D, **kw)
    batch, dummy = fe.shard_as_batch(rp, cols, vals, y, off, wt, True)
    pk = oracle.pack(batch.ent_row_ptr, batch.row_nnz_ptr, batch.col_global)
    o = oracle.make_opts(l2=2.0, regularize_bias=False, has_intercept=True, max_iter=60, threshold=0.0, sum_loss=True)
    res = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o)
    th_o = fe.to_global(res["theta"], pk["unique_global"], D, True, dummy)
    for th, info in ((th_team, info_team), (th_step, info_step)):
        assert info["status"] == res["status"][0]
        assert rel_err(th, th_o) <= tol(info["status"]), rel_err(th, th_o)


@pytest.mark.gpu
def test_one_launch_step_and_look_ahead_give_the_bits_of_the_plain_loop(device_solver, monkeypatch):
    """Round 5: the step is ONE launch (fe_tail_kernel: the products of every virtual block, the last-arriving workgroup's decision,
    the update and the shard's local copy of x) instead of three plus fe_prepare_kernel, and the host reads the status a few steps
    late (gdmix_fe_solve / run_stepping_loop; the kernels of a stopped problem return at once). Same partition of the sums, same
    order: coefficients, value, nit, nfev are bitwise those of the three-launch loop with the status read after every step — cold
    and warm-started (the local copy of the start point is written at creation), with several row / column blocks and a frequent
    column, on a shard with fewer coefficients than one workgroup has threads and on one with more virtual blocks than CUs."""
    for n, D, seed in ((3000, 150, 1), (9000, 7000, 2), (40_000, 140_000, 3)):
        rng = np.random.default_rng(seed)
        k = rng.integers(0, 24, n)
        k[rng.integers(0, n, 3)] = 2000
        rp = np.concatenate([[0], np.cumsum(k)]).astype(np.int64)
        cols = np.minimum((float(D + 1) ** rng.random(rp[-1])).astype(np.int64) - 1, D - 1)
        vals = (rng.standard_normal(rp[-1]) * 0.3).astype(np.float32)
        y = (rng.random(n) < 0.3).astype(np.float32)
        off = (0.2 * rng.standard_normal(n)).astype(np.float32)
        s = fe.FixedEffectDeviceSolver(solver=device_solver)
        kw = dict(offset=off, l2=1.5, regularize_bias=False, max_iter=30)
        runs = {}
        for fused, ahead in (("0", 0), ("1", 0), ("1", 2), ("1", 5), ("0", 3)):
            monkeypatch.setenv("GDMIX_FE_FUSED_TAIL", fused)
            monkeypatch.setattr(fe, "LOOKAHEAD", ahead)
            monkeypatch.setenv("GDMIX_FE_HOT_MIN", "300")
            th, info = s.fit_stepping(rp, cols, vals, y, D, **kw)
            th_w, info_w = s.fit_stepping(rp, cols, vals, y, D, theta0=th * 0.5, **dict(kw, max_iter=7))
            runs[(fused, ahead)] = (th, info, th_w, info_w)
        ref = runs[("0", 0)]
        assert ref[1]["nit"] > 5 and ref[3]["nit"] >= 1 and np.abs(ref[0]).max() > 0
        for key, (th, info, th_w, info_w) in runs.items():
            assert np.array_equal(th, ref[0]) and info == ref[1], key
            assert np.array_equal(th_w, ref[2]) and info_w == ref[3], key
        # ... and the Hessian passes, which run after the stop at a point of their own, leave the solver's local copy alone
        monkeypatch.setenv("GDMIX_FE_FUSED_TAIL", "1")
        th_v, info_v = s.fit_stepping(rp, cols, vals, y, D, variance_mode="simple", threshold=0.0, **kw)
        assert np.array_equal(th_v, ref[0]) and np.all(info_v["variances"] > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("compress", ["3", "0"])
@pytest.mark.parametrize("chunk,pack,hot,window", [(None, "1", None, None), ("8192", "1", None, None), ("257", "1", None, None),
                                                   ("257", "0", None, None), (None, "0", None, None), (None, "1", "300", None),
                                                   ("257", "1", "2000", None), ("257", "0", "40", None), (None, "1", None, "10"),
                                                   ("257", "1", "300", "11"), ("8192", "0", None, "12")])
def test_passes_cut_into_units_are_deterministic_and_agree(device_solver, monkeypatch, chunk, pack, hot, window, compress):
    """csrc/fe_solve.hip: several row blocks and column blocks, blocks cut into several units (GDMIX_FE_CHUNK forces that on a
    small shard; 257 is no multiple of anything), both forms of the entries (GDMIX_FE_PACK=0: the three arrays a unit spanning more
    than 2^21 gathered elements needs), a frequent feature, empty rows and features that never occur. Two fits are
    bitwise equal (one wavefront per accumulator set, in-order LDS adds); every cut agrees with the oracle. hot: GDMIX_FE_HOT_MIN
    lowers the entry count from which a column gets 32 accumulators of its own in the column pass (65 536 in production): a
    handful of columns, or the 64 most frequent of hundreds, go through that here. window: GDMIX_FE_WINDOW_BITS narrows the
    windows of gathered elements a unit may span (2^21 in production, so that the packed word always holds the key): blocks
    of this shard are cut at every 1024 / 2048 / 4096 samples or features as well."""
    if window is None:
        monkeypatch.delenv("GDMIX_FE_WINDOW_BITS", raising=False)
    else:
        monkeypatch.setenv("GDMIX_FE_WINDOW_BITS", window)
    if hot is None:
        monkeypatch.delenv("GDMIX_FE_HOT_MIN", raising=False)
    else:
        monkeypatch.setenv("GDMIX_FE_HOT_MIN", hot)
    if chunk is None:
        monkeypatch.delenv("GDMIX_FE_CHUNK", raising=False)
    else:
        monkeypatch.setenv("GDMIX_FE_CHUNK", chunk)
    monkeypatch.setenv("GDMIX_FE_PACK", pack)
    # compress: GDMIX_FE_COMPRESS, bit 0 the row pass, bit 1 the column pass — units whose key deltas fit five bits with few fillers are
    # read in the 6-byte form (values + 16-bit {delta, accumulator} words, keys rebuilt by a wavefront scan); the sparse blocks of
    # this Zipf-distributed shard keep the 8-byte form, units of 257 entries are mostly padding and keep it too: both forms in one launch
    monkeypatch.setenv("GDMIX_FE_COMPRESS", compress)
    rng = np.random.default_rng(11)
    n, D = 9000, 7000
    k = rng.integers(0, 24, n)
    k[rng.integers(0, n, 3)] = 3000
    rp = np.concatenate([[0], np.cumsum(k)]).astype(np.int64)
    cols = np.minimum((float(D + 1) ** rng.random(rp[-1])).astype(np.int64) - 1, D - 1)     # feature j with probability ~ 1/(j+1)
    vals = (rng.standard_normal(rp[-1]) * 0.3).astype(np.float32)
    y = (rng.random(n) < 0.3).astype(np.float32)
    off = (0.2 * rng.standard_normal(n)).astype(np.float32)
    wt = (0.5 + rng.random(n)).astype(np.float32)
    s = fe.FixedEffectDeviceSolver(solver=device_solver)
    kw = dict(offset=off, weight=wt, l2=1.5, regularize_bias=False, max_iter=40)
    th1, info1 = s.fit_stepping(rp, cols, vals, y, D, **kw)
    th2, info2 = s.fit_stepping(rp, cols, vals, y, D, **kw)
    assert np.array_equal(th1, th2) and info1 == info2
    batch, dummy = fe.shard_as_batch(rp, cols, vals, y, off, wt, True)
    pk = oracle.pack(batch.ent_row_ptr, batch.row_nnz_ptr, batch.col_global)
    o = oracle.make_opts(l2=1.5, regularize_bias=False, has_intercept=True, max_iter=40, threshold=0.0, sum_loss=True)
    res = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o)
    th_o = fe.to_global(res["theta"], pk["unique_global"], D, True, dummy)
    assert info1["status"] == res["status"][0] and info1["nit"] == res["nit"][0] and info1["nfev"] == res["nfev"][0]
    assert rel_err(th1, th_o) <= 1e-9, rel_err(th1, th_o)
    # the Hessian diagonal goes through the same passes
    th3, info3 = s.fit_stepping(rp, cols, vals, y, D, variance_mode="simple", threshold=0.0, **kw)
    assert np.array_equal(th3, th1)
    rows = np.repeat(np.arange(n), k)
    z = np.bincount(rows, weights=vals.astype(np.float64) * th1[cols], minlength=n) + th1[D] + off
    rho = 1 / (1 + np.exp(-z))
    d = rho * (1 - rho) * wt
    H = np.concatenate([np.bincount(cols, weights=vals.astype(np.float64) ** 2 * d[rows], minlength=D) + 1.5, [d.sum()]])
    np.testing.assert_allclose(info3["variances"], 1.0 / (H + 1e-12), rtol=1e-10)


@pytest.mark.gpu
def test_large_shard_properties(device_solver):
    """2 M samples x 16 Zipf-distributed columns of 200 k features, where the oracle takes too long for a test: properties that
    hold whatever the size. [gradient, value] of a shard is the sum of those of its two halves (what the all-reduce relies on) and
    of any other split; evaluating twice gives the same bits; a fit run twice gives the same bits."""
    rng = np.random.default_rng(21)
    n, k, D = 2_000_000, 16, 200_000
    cols = np.minimum((float(D + 1) ** rng.random((n, k))).astype(np.int64) - 1, D - 1).ravel()
    vals = (rng.standard_normal(n * k) * 0.3).astype(np.float32)
    y = (rng.random(n) < 0.35).astype(np.float32)
    off = (0.2 * rng.standard_normal(n)).astype(np.float32)
    wt = (0.5 + rng.random(n)).astype(np.float32)
    th0 = 0.05 * rng.standard_normal(D + 1)
    opts = SolverOptions(l2=2.0, regularize_bias=False, has_intercept=True, m=10, max_iter=8, threshold=0.0, sum_loss=True)
    t0 = device_solver.torch.from_numpy(th0).to(device_solver.device)

    def local_sums(rows):
        r0, r1 = rows
        rp = np.arange(r1 - r0 + 1, dtype=np.int64) * k
        batch, _ = fe.shard_as_batch(rp, cols[r0 * k:r1 * k], vals[r0 * k:r1 * k], y[r0:r1], off[r0:r1], wt[r0:r1], True, dummy=False)
        prob = fe._SteppingProblem(device_solver, device_solver.pack(batch, has_intercept=True), D, opts, t0)
        prob.eval()
        a = prob.reduce_tensor().cpu().numpy().copy()
        prob.eval()
        assert np.array_equal(prob.reduce_tensor().cpu().numpy(), a)
        prob.close()
        return a

    whole = local_sums((0, n))
    assert np.isfinite(whole).all() and np.abs(whole[:D]).max() > 0
    for cut in (n // 2, 123_457):
        parts = local_sums((0, cut)) + local_sums((cut, n))
        np.testing.assert_allclose(parts, whole, rtol=1e-11, atol=1e-9 * np.abs(whole).max())
    s = fe.FixedEffectDeviceSolver(solver=device_solver)
    rp = np.arange(n + 1, dtype=np.int64) * k
    fits = [s.fit_stepping(rp, cols, vals, y, D, offset=off, weight=wt, l2=2.0, regularize_bias=False, max_iter=8, theta0=th0) for _ in range(2)]
    assert np.array_equal(fits[0][0], fits[1][0]) and fits[0][1] == fits[1][1] and fits[0][1]["nit"] == 8


@pytest.mark.gpu
def test_two_workers_all_reduce_gradient_and_value(tmp_path):
    """Two processes, each with every other sample as its shard: the replicated L-BFGS step on the all-reduced [gradient, value]
    gives every worker the coefficients of the whole data set. With two GPUs visible each rank has its own device and the
    all-reduce is RCCL in place on the device buffer; on the 1-GPU box both sit on GPU 0 and the collective is gloo."""
    import json
    import subprocess
    import sys
    names = ["logistic_offset", "linear_offset", "logistic_wide", "logistic_no_intercept", "logistic_warm_one_iteration"]
    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TF_CONFIG", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(root, "tests", "_fe_dist_worker.py"), str(tmp_path), ",".join(names)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=root)
    res = json.load(open(tmp_path / "result.json"))
    assert len(res) == 2
    import torch
    want = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    assert res[0]["_backend"] == res[1]["_backend"] == want
    assert (res[0]["_device"], res[1]["_device"]) == ((0, 1) if want == "nccl" else (0, 0))
    for name in names:
        c = load(name)
        a, b = res[0][name], res[1][name]
        assert a["theta"] == b["theta"] and a["status"] == b["status"] and a["nit"] == b["nit"]     # replicated step: bitwise equal
        assert rel_err(np.array(a["theta"]), c["theta"]) <= tol(a["status"]) * 10, (name, rel_err(np.array(a["theta"]), c["theta"]))


@pytest.mark.gpu
def test_two_workers_full_variances_on_the_device(tmp_path):
    """fixed_effect_variance_mode FULL with two workers and a model wider than the host path takes (4 501 coefficients): each
    worker's dense curvature matrix on its device, one all-reduce, the factorisation replicated — against numpy's inverse of the
    whole data set's Hessian at the returned coefficients (fixed_effect_lr_lbfgs_model.py:291-305, 384-389, 457-463)."""
    import json
    import subprocess
    import sys
    import scipy.sparse as sp
    root = os.path.dirname(HERE)
    sys.path.insert(0, os.path.join(root, "tests"))
    from _fe_dist_worker import wide_case
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TF_CONFIG", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29619", os.path.join(root, "tests", "_fe_dist_worker.py"), str(tmp_path), "logistic_offset", "full_variance"]
    subprocess.run(cmd, check=True, env=env, timeout=900, cwd=root)
    res = json.load(open(tmp_path / "result.json"))
    a, b = res[0]["full_variance"], res[1]["full_variance"]
    assert a["theta"] == b["theta"] and a["variances"] == b["variances"]             # replicated: bitwise equal on both workers
    rp, col, val, y, off, wt, D = wide_case()
    theta = np.array(a["theta"])
    X = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(rp.size - 1, D))
    X = sp.hstack([X, sp.csr_matrix(np.ones((rp.size - 1, 1)))], format="csr")
    rho = 1.0 / (1.0 + np.exp(-(X @ theta + off)))
    H = np.asarray((X.T @ X.multiply((rho * (1 - rho) * wt)[:, None])).todense())
    H += np.diag([0.7 + 1e-12] * (D + 1))
    H[-1, -1] -= 0.7                                                                   # the intercept is not regularised
    want = np.diagonal(np.linalg.inv(H))
    got = np.array(a["variances"])
    assert got.shape == (D + 1,)
    np.testing.assert_allclose(got, want, rtol=1e-8)
    np.testing.assert_allclose(got[D - 40:D], 1.0 / (0.7 + 1e-12), rtol=1e-12)       # features no worker has seen


@pytest.mark.gpu
def test_device_fixed_effect_at_scale_against_oracle(device_solver):
    """600k samples x 8 non-zeros over 5000 features, weights, unregularised intercept, both model types."""
    rng = np.random.default_rng(0)
    n, k, D = 600_000, 8, 5000   # 293 row blocks, 3 column blocks cut into units
    cols = rng.integers(0, D, (n, k))
    vals = rng.standard_normal((n, k)).astype(np.float32)
    w_true = rng.standard_normal(D) * 0.3
    z = (vals * w_true[cols]).sum(1) + 0.2
    off = (0.1 * rng.standard_normal(n)).astype(np.float32)
    wt = (0.5 + rng.random(n)).astype(np.float32)
    rp = np.arange(n + 1, dtype=np.int64) * k
    for linear in (False, True):
        y = (z + 0.1 * rng.standard_normal(n)).astype(np.float32) if linear else (rng.random(n) < 1 / (1 + np.exp(-z))).astype(np.float32)
        s = fe.FixedEffectDeviceSolver(solver=device_solver)
        theta, info = s.fit(rp, cols.ravel(), vals.ravel(), y, D, offset=off, weight=wt, l2=10.0, regularize_bias=False,
                            model_type=fe.LINEAR_REGRESSION if linear else fe.LOGISTIC_REGRESSION, max_iter=200)
        batch, dummy = fe.shard_as_batch(rp, cols.ravel(), vals.ravel(), y, off, wt, True, binary_labels=not linear)
        pk = oracle.pack(batch.ent_row_ptr, batch.row_nnz_ptr, batch.col_global)
        o = oracle.make_opts(l2=10.0, regularize_bias=False, has_intercept=True, max_iter=200, threshold=0.0, sum_loss=True, linear=linear)
        res = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o)
        th_o = fe.to_global(res["theta"], pk["unique_global"], D, True, dummy)
        assert info["status"] in (0, 1) and res["status"][0] == info["status"]
        assert rel_err(theta, th_o) <= 1e-5, rel_err(theta, th_o)
        th2, info2 = s.fit_stepping(rp, cols.ravel(), vals.ravel(), y, D, offset=off, weight=wt, l2=10.0, regularize_bias=False,
                                    model_type=fe.LINEAR_REGRESSION if linear else fe.LOGISTIC_REGRESSION, max_iter=200)
        assert info2["status"] == info["status"]
        assert rel_err(th2, th_o) <= 1e-5, rel_err(th2, th_o)


@pytest.mark.gpu
def test_a_gradient_that_falls_by_thousands_inside_the_history_window(device_solver):
    """tools/fuzz_fe.py case 6700230 (round 6): squared loss, 287 742 samples x 2 non-zeros over 150 000 features, weights, offsets,
    l2 = 10; |g| goes from 3 000 to 0.9 in twelve iterations and the thirteenth amplifies a perturbation 10^4 times. scipy's
    L-BFGS-B and the oracle take 25 iterations. With S'g and Y'g kept as running sums of the products with y (rounds 3 - 5) the
    device left their trajectory by 1e-5 at iteration 13 and stopped after 47 (stepping kernels) / 36 (one kernel) iterations,
    2.5e-4 away; with the products taken directly it stays within the oracle's own sensitivity (profiles/r06_fuzz.txt)."""
    c = fuzz_fe_case.draw(6700230)
    n, D, rp, cols, vals, y, off, wt = c.n, c.D, c.rp, c.cols, c.vals, c.y, c.off, c.wt
    assert (n, D, c.Z, c.linear, c.ic, c.l2, c.regb, c.max_iter, c.m) == (287742, 150000, 574536, True, True, 10.0, False, 200, 10) and c.th0 is None
    kw = dict(offset=off, weight=wt, has_intercept=True, l2=10.0, regularize_bias=False, model_type=fe.LINEAR_REGRESSION, max_iter=200, m=10)
    batch, dummy = fe.shard_as_batch(rp, cols, vals, y, off, wt, True, binary_labels=False)
    pk = oracle.pack(batch.ent_row_ptr, batch.row_nnz_ptr, batch.col_global)
    o = oracle.make_opts(l2=10.0, regularize_bias=False, has_intercept=True, m=10, max_iter=200, threshold=0.0, sum_loss=True, linear=True)
    res = oracle.solve(pk, batch.val, batch.y, batch.offset, batch.weight, o)
    th_o = fe.to_global(res["theta"], pk["unique_global"], D, True, dummy)
    assert int(res["status"][0]) == 1 and int(res["nit"][0]) == 25 and int(res["nfev"][0]) == 35      # what scipy 1.15's L-BFGS-B does on this objective
    s = fe.FixedEffectDeviceSolver(solver=device_solver)
    for fit in (s.fit_stepping, s.fit):
        th, info = fit(rp, cols, vals, y, D, **kw)
        assert (int(info["status"]), int(info["nit"]), int(info["nfev"])) == (1, 25, 35), (fit.__name__, info)
        assert rel_err(th, th_o) <= REL_TOL_FACTR, (fit.__name__, rel_err(th, th_o))   # observed 2.5e-7 (the oracle under 1e-13 start noise: 5e-8; scipy on a numpy objective: 6e-6); the sums: 2.5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("has_intercept", [True, False])
def test_device_scoring_of_a_raw_shard(device_solver, has_intercept):
    """gdmix_fe_score: one pass over the sample-major arrays (ragged rows, empty rows, no pack) equals the host sum, and the
    packed scoring pass it replaced, to float rounding."""
    rng = np.random.default_rng(9)
    n, D = 5000, 3000
    k = rng.integers(0, 30, n)
    k[:2] = 0
    k[-1] = 0
    k[7] = 2500
    rp = np.concatenate([[0], np.cumsum(k)]).astype(np.int64)
    cols = rng.integers(0, D, rp[-1])
    vals = rng.standard_normal(rp[-1]).astype(np.float32)
    off = rng.standard_normal(n).astype(np.float32)
    theta = rng.standard_normal(D + (1 if has_intercept else 0))
    s = fe.FixedEffectDeviceSolver(solver=device_solver)
    score, per = s.score(rp, cols, vals, off, theta, D, has_intercept)
    want = np.zeros(n)
    np.add.at(want, np.repeat(np.arange(n), k), vals.astype(np.float64) * theta[cols])
    if has_intercept:
        want += theta[D]
    np.testing.assert_allclose(per, want.astype(np.float32), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(score, (want + off).astype(np.float32), rtol=2e-5, atol=2e-5)
    # the path it replaced: pack the shard as one entity and score in the local index space
    batch, dummy = fe.shard_as_batch(rp, cols, vals, np.zeros(n, np.float32), off, None, has_intercept)
    packed = device_solver.pack(batch, has_intercept=has_intercept)
    local = fe.to_local(theta, packed.unique_global().cpu().numpy(), D, has_intercept, dummy)
    lo, pc = device_solver.score(packed, local)
    np.testing.assert_array_equal(score, lo.cpu().numpy())
    np.testing.assert_array_equal(per, pc.cpu().numpy())
    # no feature bag (intercept-only model), no offset
    if has_intercept:
        sc, pe = s.score(None, None, None, off, theta[D:], 0, True)
        np.testing.assert_allclose(pe, np.full(n, theta[D], np.float32), rtol=1e-7)
    with pytest.raises(ValueError):
        s.score(rp, np.where(cols == cols[0], D, cols), vals, off, theta, D, has_intercept)


@pytest.mark.gpu
@pytest.mark.parametrize("regularize_bias", [False, True])
def test_device_hessian_diagonal_and_simple_variance(device_solver, regularize_bias):
    """gdmix_fe_hessian_diag: two streaming passes (rows -> w rho (1 - rho), columns -> sum val^2 d) against dense numpy, ragged
    rows, weights, 293 row blocks; then the SIMPLE variances through fit_stepping."""
    rng = np.random.default_rng(17)
    n, D = 600_000, 700
    k = rng.integers(0, 9, n)
    k[:3] = 0
    rp = np.concatenate([[0], np.cumsum(k)]).astype(np.int64)
    cols = np.concatenate([rng.choice(D, kk, replace=False) for kk in k[:2000]] + [rng.integers(0, D, int(k[2000:].sum()))]) if True else None
    # (distinct columns inside a row for the first rows; the bulk may repeat a column in a row: both sides then add val^2 per entry)
    vals = rng.standard_normal(rp[-1]).astype(np.float32)
    y = (rng.random(n) < 0.5).astype(np.float32)
    off = (0.3 * rng.standard_normal(n)).astype(np.float32)
    wt = (0.5 + rng.random(n)).astype(np.float32)
    s = fe.FixedEffectDeviceSolver(solver=device_solver)
    l2 = 3.0
    theta, info = s.fit_stepping(rp, cols, vals, y, D, offset=off, weight=wt, l2=l2, regularize_bias=regularize_bias, max_iter=30,
                                 variance_mode="simple", threshold=1e-4)
    th = np.where(np.abs(theta) <= 1e-4, 0.0, theta)
    rows = np.repeat(np.arange(n), k)
    z = np.bincount(rows, weights=vals.astype(np.float64) * th[cols], minlength=n) + th[D] + off
    rho = 1 / (1 + np.exp(-z))
    d = rho * (1 - rho) * wt
    H = np.concatenate([np.bincount(cols, weights=vals.astype(np.float64) ** 2 * d[rows], minlength=D), [d.sum()]]) + l2
    if not regularize_bias:
        H[-1] -= l2
    np.testing.assert_allclose(info["variances"], 1.0 / (H + 1e-12), rtol=1e-10)


def test_variance_request_is_checked_before_training():
    """fixed_effect_variance_mode FULL beyond what can be inverted fails before any device work, not after the L-BFGS loop
    (fixed_effect_lr_lbfgs_model.py:291,457 has no limit but memory; here: 16 384 coefficients, on the device above 4 096,
    with any number of workers)."""
    fe.check_variance_request("SIMPLE", 10 ** 7, 8)
    fe.check_variance_request("FULL", fe.FULL_VARIANCE_HOST_MAX, 8)
    fe.check_variance_request("FULL", fe.FULL_VARIANCE_DEVICE_MAX, 1)
    fe.check_variance_request("FULL", fe.FULL_VARIANCE_DEVICE_MAX, 8)     # several workers: the P x P matrix is all-reduced (round 4)
    with pytest.raises(ValueError, match="at most"):
        fe.check_variance_request("FULL", fe.FULL_VARIANCE_DEVICE_MAX + 1, 1)
    with pytest.raises(ValueError, match="unknown variance mode"):
        fe.check_variance_request("DIAGONAL", 10, 1)

    class NeverTouched:   # a solver whose pack() would be the first device work
        def pack(self, *a, **k):
            raise AssertionError("device work started before the variance request was checked")
    s = fe.FixedEffectDeviceSolver.__new__(fe.FixedEffectDeviceSolver)
    s.solver = NeverTouched()
    rp = np.arange(4, dtype=np.int64)
    with pytest.raises(ValueError, match="at most"):
        s.fit_stepping(rp, np.array([0, 5, 20000]), np.ones(3, np.float32), np.array([0, 1, 1], np.float32), 20001, variance_mode="FULL")


@pytest.mark.gpu
@pytest.mark.parametrize("regularize_bias", [False, True])
def test_device_full_variance_beyond_the_host_limit(device_solver, regularize_bias):
    """FULL variances of a model of 5 001 coefficients (above the 4 096 the host path takes): Hessian, tiled Cholesky and inverse on
    the device, against the dense statement the reference evaluates (fixed_effect_lr_lbfgs_model.py:296-305, 457-463) in numpy.
    Some features never occur in the shard: their variance is 1 / (l2 + 1e-12)."""
    rng = np.random.default_rng(23)
    n, k, D = 30_000, 6, 5000
    cols = rng.integers(0, D - 40, (n, k))          # the last 40 features have no non-zero
    vals = rng.standard_normal((n, k)).astype(np.float32)
    y = (rng.random(n) < 0.45).astype(np.float32)
    off = (0.2 * rng.standard_normal(n)).astype(np.float32)
    wt = (0.5 + rng.random(n)).astype(np.float32)
    rp = np.arange(n + 1, dtype=np.int64) * k
    s = fe.FixedEffectDeviceSolver(solver=device_solver)
    l2 = 2.5
    theta, info = s.fit_stepping(rp, cols.ravel(), vals.ravel(), y, D, offset=off, weight=wt, l2=l2, regularize_bias=regularize_bias,
                                 max_iter=15, variance_mode="full", threshold=1e-4)
    th = np.where(np.abs(theta) <= 1e-4, 0.0, theta)
    X = np.zeros((n, D + 1))
    np.add.at(X, (np.repeat(np.arange(n), k), cols.ravel()), vals.ravel().astype(np.float64))
    X[:, D] = 1.0
    rho = 1 / (1 + np.exp(-(X @ th + off)))
    H = (X * (rho * (1 - rho) * wt)[:, None]).T @ X + (l2 + 1e-12) * np.eye(D + 1)
    if not regularize_bias:
        H[D, D] -= l2
    want = np.diag(np.linalg.inv(H))
    np.testing.assert_allclose(info["variances"], want, rtol=1e-8)
    np.testing.assert_allclose(info["variances"][D - 40:D], 1.0 / (l2 + 1e-12), rtol=1e-12)


# ---- the stepping loop's failure protocol, on stand-ins: no device ------------------------------------------------------

class _StubProblem:
    """What run_stepping_loop drives: statuses[k] is the status of step k (-1 = go on)."""

    def __init__(self, statuses):
        self.statuses, self.evals, self.steps = list(statuses), 0, 0

    def reduce_tensor(self):
        return "buf"

    def eval(self):
        self.evals += 1

    def step_async(self):
        self.steps += 1
        return self.steps - 1

    def step_status(self, seq):
        return self.statuses[min(seq, len(self.statuses) - 1)]


@pytest.mark.parametrize("lookahead", [0, 2])
def test_every_worker_of_an_aborted_fit_enqueues_the_same_number_of_all_reduces(lookahead):
    """Step k of worker A times out (status 9). Its value slot carries the mark, the NEXT all-reduce hands it to worker B, whose step
    k + 1 stops with status 10. A joins one more all-reduce before it raises; B none: both have enqueued k + lookahead + 2, nobody is
    left alone in a collective."""
    from gdmix_amd import fixed_effect as fe
    k = 3
    counts = {}
    for name, statuses in (("A", [-1] * k + [fe.ST_ABORTED]), ("B", [-1] * (k + 1) + [fe.ST_ABORTED_PEER])):
        n = {"all_reduce": 0}

        def all_reduce(buf):
            n["all_reduce"] += 1
        with pytest.raises(RuntimeError, match="aborted|gave up"):
            fe.run_stepping_loop(_StubProblem(statuses), all_reduce=all_reduce, lookahead=lookahead)
        counts[name] = n["all_reduce"]
    assert counts["A"] == counts["B"] == k + lookahead + 2


def test_a_lookahead_the_status_ring_cannot_hold_is_refused_before_the_first_collective():
    from gdmix_amd import fixed_effect as fe
    n = {"all_reduce": 0}
    with pytest.raises(ValueError, match="lookahead"):
        fe.run_stepping_loop(_StubProblem([0]), all_reduce=lambda b: n.__setitem__("all_reduce", n["all_reduce"] + 1), lookahead=fe.FE_RING)
    assert n["all_reduce"] == 0
    assert fe.run_stepping_loop(_StubProblem([-1, -1, 1]), all_reduce=lambda b: None, lookahead=fe.FE_RING - 1) == 1
