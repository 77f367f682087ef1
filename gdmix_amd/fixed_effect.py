"""Fixed-effect linear / logistic regression on one MI355X (single-worker form).

The reference trains the fixed effect by scipy's fmin_l_bfgs_b over an objective TensorFlow evaluates on the
worker's shard (gdmix-trainer/src/gdmix/models/custom/fixed_effect_lr_lbfgs_model.py:309-392,635-643):

    f(theta) = sum_i w_i * loss(y_i, x_i . w + b + offset_i) + (l2 / 2) * |theta_reg|^2        (not divided by n)
    loss     = sigmoid cross entropy (logistic_regression) or (y - z)^2 (linear_regression)
    theta    = [w (num_features), b]: the intercept is the LAST coefficient (:340-342)
    theta_reg = theta if regularize_bias or no intercept, else theta[:-1] (:367-369)

That is the random-effect objective of one "entity" holding the whole shard, without the 1/n and with an optional
squared loss — two switches of the same device solver (gdmix_re_opts.sum_loss / .linear, include/gdmix_re.h): the
shard is packed as a one-entity batch and solved by the device-wide team kernel (csrc/re_solve_team.hpp), X
streaming from HBM twice per evaluation (CSR for the logits, the CSC copy for the gradient). This module maps
between the reference's coefficient layout (global feature index, intercept last) and the solver's (features
present in the shard, intercept first).

Two device paths, same arithmetic rules:
  * fit()           the whole solve inside the device-wide team kernel (one launch; single worker only);
  * fit_stepping()  include/gdmix_fe.h: streaming kernels for the two passes over the shard and the L-BFGS step as
                    separate launches, with ONE all-reduce of [gradient, value] across workers in between (the
                    reference does two, :375-381) — RCCL through torch.distributed on the GPU box. Every worker holds
                    its shard and runs the replicated step on the reduced buffer, like the reference's workers run scipy.
"""
import ctypes as C
import os

import numpy as np

from .batch import RawBatch
from .solver import REDeviceSolver, SolverOptions

LOGISTIC_REGRESSION = "logistic_regression"
LINEAR_REGRESSION = "linear_regression"


def shard_as_batch(row_nnz_ptr, col_global, val, y, offset=None, weight=None, has_intercept=True, binary_labels=True,
                   dummy=None):
    """The shard as a one-entity RawBatch.

    dummy=True: an intercept-only model (no feature bag configured) gets one dummy zero feature per sample, as the
    random-effect path does (job_consumers.py:213-218). dummy=None infers it from "no non-zeros at all", which is
    only right for callers that have no notion of a feature bag; the model class passes `feature_bag is None`.

    A bagged worker whose shard holds no non-zero (no samples at all — shard_input_files hands out [] when there are
    fewer files than workers, util/distribution_utils.py:41-44 — or only empty rows) still owns the full
    coefficient space: it gets one extra sample of weight 0 carrying a 0.0 at feature 0, which adds exactly 0 to the
    value and to every gradient entry, so it joins the all-reduce with a full-size buffer of zeros like the
    reference's empty worker does."""
    row_nnz_ptr = np.asarray(row_nnz_ptr, np.int64)
    n = row_nnz_ptr.size - 1
    col = np.asarray(col_global, np.int64)
    v = np.asarray(val, np.float32)
    y = np.asarray(y, np.float32)
    offset = np.zeros(n, np.float32) if offset is None else np.asarray(offset, np.float32)
    weight = None if weight is None else np.asarray(weight, np.float32)
    if dummy is None:
        dummy = col.size == 0
    if dummy:
        if not has_intercept:
            raise ValueError("a model without features needs an intercept")
        row_nnz_ptr = np.arange(n + 1, dtype=np.int64)
        col = np.zeros(n, np.int64)
        v = np.zeros(n, np.float32)
    if col.size == 0:   # bagged shard without a single non-zero (or a dummy shard without samples): the weight-0 sample
        row_nnz_ptr = np.concatenate([row_nnz_ptr, [row_nnz_ptr[-1] + 1]]).astype(np.int64)
        col = np.zeros(1, np.int64)
        v = np.zeros(1, np.float32)
        y = np.concatenate([y, np.zeros(1, np.float32)])
        offset = np.concatenate([offset, np.zeros(1, np.float32)])
        weight = np.concatenate([np.ones(n, np.float32) if weight is None else weight, np.zeros(1, np.float32)])
        n += 1
    b = RawBatch(ent_row_ptr=np.array([0, n], np.int64), row_nnz_ptr=row_nnz_ptr, col_global=col, val=v,
                 y=y, offset=offset, weight=weight, uid=np.arange(n, dtype=np.int64),
                 entity_ids=["fixed_effect"], has_label=True, binary_labels=binary_labels)
    return b, dummy


def to_local(theta_global, unique_global, num_features, has_intercept, dummy):
    """reference layout [w (num_features), b] -> solver layout [b, w of the features present]."""
    t = np.asarray(theta_global, np.float64)
    ic = 1 if has_intercept else 0
    out = np.zeros(unique_global.size + ic)
    if has_intercept:
        out[0] = t[num_features]
    if not dummy:
        out[ic:] = t[unique_global]
    return out


def to_global(theta_local, unique_global, num_features, has_intercept, dummy):
    ic = 1 if has_intercept else 0
    out = np.zeros(num_features + ic)
    if not dummy:
        out[unique_global] = theta_local[ic:]
    if has_intercept:
        out[num_features] = theta_local[0]
    return out


class FixedEffectDeviceSolver:
    """fit() = FixedEffectLRModelLBFGS's training step for one worker, on the device."""

    def __init__(self, device=0, solver=None):
        self.solver = solver or REDeviceSolver(device)

    def score(self, row_nnz_ptr, col_global, val, offset, theta, num_features, has_intercept=True):
        """-> (score, per-coordinate score) of every sample under theta (intercept last); see device_score."""
        return device_score(self.solver, row_nnz_ptr, col_global, val, offset, theta, num_features, has_intercept)

    def fit(self, row_nnz_ptr, col_global, val, y, num_features, offset=None, weight=None, has_intercept=True, l2=1.0,
            regularize_bias=True, model_type=LOGISTIC_REGRESSION, theta0=None, max_iter=100, m=10, tolerance=1e-12, dummy=None):
        """-> (theta [num_features + has_intercept], intercept last; info dict with f, nit, nfev, status, gnorm)."""
        if model_type not in (LOGISTIC_REGRESSION, LINEAR_REGRESSION):
            raise ValueError(f"unknown model type {model_type!r}")
        batch, dummy = shard_as_batch(row_nnz_ptr, col_global, val, y, offset, weight, has_intercept,
                                      binary_labels=(model_type == LOGISTIC_REGRESSION), dummy=dummy)
        if not dummy and batch.col_global.size and (batch.col_global.min() < 0 or batch.col_global.max() >= num_features):
            raise ValueError(f"feature index outside [0, {num_features})")
        packed = self.solver.pack(batch, has_intercept=has_intercept)
        uniq = packed.unique_global().cpu().numpy()
        opts = SolverOptions(l2=l2, regularize_bias=bool(regularize_bias) and bool(has_intercept), has_intercept=has_intercept, m=m,
                             max_iter=max_iter, ftol=tolerance, threshold=0.0, sum_loss=True,
                             linear=(model_type == LINEAR_REGRESSION))
        if not has_intercept:
            opts.regularize_bias = False
        t0 = None if theta0 is None else to_local(theta0, uniq, num_features, has_intercept, dummy)
        res = self.solver.solve(packed, opts, theta0=t0).to_host()
        theta = to_global(res["theta"], uniq, num_features, has_intercept, dummy)
        info = {k: res[k][0] for k in ("fval", "nit", "nfev", "status", "gnorm")}
        if not 0 <= int(info["status"]) <= 4:
            raise RuntimeError(f"the fixed-effect solve did not finish (device status {int(info['status'])}: 9 = a device barrier timed out)")
        return theta, info


def device_score(solver, row_nnz_ptr, col_global, val, offset, theta, num_features, has_intercept):
    """(score, per-coordinate score) float32 numpy arrays of every sample: x . w + b (+ offset). theta: [num_features +
    has_intercept], intercept last. The shard is read once off its sample-major arrays (gdmix_fe_score): no pack."""
    t = solver.torch
    n = (len(row_nnz_ptr) - 1) if row_nnz_ptr is not None else len(offset)
    dev = solver.device
    up = lambda a, dt: None if a is None else t.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    rp, cg, vl = up(row_nnz_ptr, np.int64), up(col_global, np.int64), up(val, np.float32)
    if cg is not None and cg.numel() and (int(cg.min()) < 0 or int(cg.max()) >= num_features):
        raise ValueError(f"feature index outside [0, {num_features})")
    of = up(offset, np.float32)
    th = up(theta, np.float64)
    assert th.numel() == num_features + (1 if has_intercept else 0)
    score = t.empty(n, dtype=t.float32, device=dev)
    per = t.empty(n, dtype=t.float32, device=dev)
    ptr = lambda x: None if x is None or x.numel() == 0 else x.data_ptr()
    rc = solver.lib.gdmix_fe_score(solver._h, n, None if cg is None or cg.numel() == 0 else rp.data_ptr(), ptr(cg), ptr(vl), ptr(of),
                                   th.data_ptr(), int(num_features), int(bool(has_intercept)), ptr(score), ptr(per), solver._stream())
    if rc != 0:
        from .solver import GdmixReError
        raise GdmixReError("gdmix_fe_score: " + solver.lib.gdmix_re_last_error().decode())
    return score.cpu().numpy(), per.cpu().numpy()


ST_ABORTED = 9      # GDMIX_RE_ST_ABORTED
ST_ABORTED_PEER = 10   # GDMIX_RE_ST_ABORTED_PEER: another worker's step was aborted (its mark came with the all-reduce)
FE_RING = 8         # status ring of the library (csrc/fe_solve.hip: FE_RING): a status can be read at most FE_RING - 1 steps late
LOOKAHEAD = int(os.environ.get("GDMIX_FE_LOOKAHEAD", "2"))   # evaluations enqueued ahead of the status the host has read


class _SteppingProblem:
    """gdmix_fe_problem handle; the packed shard and theta0 are kept alive with it."""

    def __init__(self, solver, packed, num_features, opts, theta0_dev):
        from .solver import GdmixReError
        self.solver, self.packed, self.theta0 = solver, packed, theta0_dev
        self.lib = solver.lib
        self._h = C.c_void_p()
        c_opts = opts.to_c()
        rc = self.lib.gdmix_fe_create(solver._h, C.byref(packed.c), int(num_features), C.byref(c_opts),
                                      None if theta0_dev is None else theta0_dev.data_ptr(), C.byref(self._h), solver._stream())
        if rc != 0:
            raise GdmixReError("gdmix_fe_create: " + self.lib.gdmix_re_last_error().decode())
        cnt = C.c_int64()
        ptr = self.lib.gdmix_fe_reduce_buffer(self._h, C.byref(cnt))
        self.count = int(cnt.value)
        self.buffer_ptr = ptr

    def _check(self, rc, what):
        from .solver import GdmixReError
        if rc != 0:
            raise GdmixReError(f"{what}: " + self.lib.gdmix_re_last_error().decode())

    def reduce_tensor(self):
        """The [gradient, value] buffer as a torch tensor view (no copy), for torch.distributed.all_reduce."""
        t = self.solver.torch
        return _device_view(t, self.buffer_ptr, self.count, self.solver.device)

    def eval(self):
        self._check(self.lib.gdmix_fe_eval(self._h, self.solver._stream()), "gdmix_fe_eval")

    def step(self):
        st = C.c_int32(-1)
        self._check(self.lib.gdmix_fe_step(self._h, self.solver._stream(), C.byref(st)), "gdmix_fe_step")
        return int(st.value)

    def step_async(self):
        """Enqueue the step; returns its number (gdmix_fe_step_async)."""
        seq = C.c_int64(-1)
        self._check(self.lib.gdmix_fe_step_async(self._h, self.solver._stream(), C.byref(seq)), "gdmix_fe_step_async")
        return int(seq.value)

    def step_status(self, seq):
        st = C.c_int32(-1)
        self._check(self.lib.gdmix_fe_step_status(self._h, int(seq), C.byref(st)), "gdmix_fe_step_status")
        return int(st.value)

    def solve(self, lookahead=LOOKAHEAD, max_evals=100000):
        """The whole single-worker loop inside the library (gdmix_fe_solve): the status is read `lookahead` steps late."""
        st, n = C.c_int32(-1), C.c_int64(0)
        self._check(self.lib.gdmix_fe_solve(self._h, self.solver._stream(), int(lookahead), int(max_evals), C.byref(st), C.byref(n)),
                    "gdmix_fe_solve")
        return int(st.value), int(n.value)

    def result(self):
        t = self.solver.torch
        theta = t.empty(self.count - 1, dtype=t.float64, device=self.solver.device)
        f, g, nit, nfev = C.c_double(), C.c_double(), C.c_int32(), C.c_int32()
        self._check(self.lib.gdmix_fe_result(self._h, theta.data_ptr(), C.byref(f), C.byref(g), C.byref(nit), C.byref(nfev),
                                             self.solver._stream()), "gdmix_fe_result")
        return theta.cpu().numpy(), dict(fval=f.value, gnorm=g.value, nit=int(nit.value), nfev=int(nfev.value))

    def hessian_diag(self, theta_dev=None):
        """diag(X~' D X~) of the shard at theta (device tensor [D + has_intercept], None = the current point) into the reduce
        buffer (gdmix_fe_hessian_diag)."""
        self._check(self.lib.gdmix_fe_hessian_diag(self._h, None if theta_dev is None else theta_dev.data_ptr(), self.solver._stream()),
                    "gdmix_fe_hessian_diag")

    def stream_bytes(self):
        """(bytes of entries the row pass reads, the column pass reads) — gdmix_fe_stream_bytes."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self.lib.gdmix_fe_stream_bytes(self._h, C.byref(a), C.byref(b)), "gdmix_fe_stream_bytes")
        return int(a.value), int(b.value)

    def last_eval_ms(self):
        a, b = C.c_float(), C.c_float()
        self._check(self.lib.gdmix_fe_last_eval_ms(self._h, C.byref(a), C.byref(b)), "gdmix_fe_last_eval_ms")
        return float(a.value), float(b.value)

    def close(self):
        if self._h:
            self.lib.gdmix_fe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _device_view(torch, ptr, count, device):
    """float64 tensor over `count` doubles of device memory owned by the library."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)


def run_stepping_loop(problem, all_reduce=None, max_evals=100000, lookahead=None):
    """do { eval; all_reduce; } while (step() < 0)  — the loop gdmix_fe.h describes, with the status read `lookahead` steps late
    so that the device never waits for the host between evaluations (the kernels of a stopped problem return at once; every
    worker takes the same decisions, so all of them enqueue the same number of all-reduces). all_reduce(tensor) sums in place
    across workers (None: single worker, the loop runs inside the library). A collective that goes through the host (gloo) is
    synchronous anyway: lookahead 0 unless all_reduce.device_ordered is set (RCCL on the device buffer)."""
    if lookahead is None:
        lookahead = LOOKAHEAD if (all_reduce is None or getattr(all_reduce, "device_ordered", False)) else 0
    if not 0 <= int(lookahead) < FE_RING:      # the status ring of the library (gdmix_fe_step_status would fail mid-run, collectives queued)
        raise ValueError(f"lookahead must be in 0..{FE_RING - 1}, not {lookahead}")
    lookahead = int(lookahead)
    if all_reduce is None:
        status, _ = problem.solve(lookahead, max_evals)
        if status in (ST_ABORTED, ST_ABORTED_PEER):
            raise RuntimeError(f"the fixed-effect step gave up waiting for a workgroup of its own launch (status {status}): the result is invalid")
        if status >= 0:
            return status
        raise RuntimeError(f"the fixed-effect L-BFGS loop did not stop within {max_evals} evaluations")
    buf = problem.reduce_tensor()
    for k in range(max_evals + lookahead):
        problem.eval()
        all_reduce(buf)
        seq = problem.step_async()
        if k >= lookahead:
            status = problem.step_status(seq - lookahead)
            if status == ST_ABORTED:
                # The abort is a per-device timeout: the other workers have not seen it. This worker's value slot carries the mark
                # (csrc/fe_solve.hip), the next all-reduce hands it to everybody and their step stops with the same status — one
                # evaluation later, so this worker joins one more all-reduce before it raises: every worker has then enqueued the
                # same number of collectives and none hangs in RCCL.
                problem.eval()
                all_reduce(buf)
                raise RuntimeError("the fixed-effect step gave up waiting for a workgroup of its own launch (status 9): the result is invalid "
                                   "(every worker stops in the same evaluation)")
            if status == ST_ABORTED_PEER:      # one evaluation behind the worker it happened on: the collectives are even, nothing to add
                raise RuntimeError("another worker's fixed-effect step was aborted (status 10): the result is invalid")
            if status >= 0:
                return problem.step_status(seq)      # (the steps behind the stop are no-ops with the same status: nothing left in flight)
    raise RuntimeError(f"the fixed-effect L-BFGS loop did not stop within {max_evals} evaluations")


def _fit_stepping(self, row_nnz_ptr, col_global, val, y, num_features, offset=None, weight=None, has_intercept=True, l2=1.0,
                  regularize_bias=True, model_type=LOGISTIC_REGRESSION, theta0=None, max_iter=100, m=10, tolerance=1e-12,
                  group=None, return_problem=False, dummy=None, variance_mode=None, threshold=0.0):
    """Same contract as fit(), through include/gdmix_fe.h. With torch.distributed initialised (or `group` given) every
    worker calls this with its own shard; the coefficients returned are identical on all workers. dummy: True for a
    model without a feature bag (intercept only), False for a bagged model — also when this worker's shard happens to
    hold no non-zero, so that its all-reduce buffer has the same num_features + 2 entries as everyone else's."""
    if model_type not in (LOGISTIC_REGRESSION, LINEAR_REGRESSION):
        raise ValueError(f"unknown model type {model_type!r}")
    batch, dummy = shard_as_batch(row_nnz_ptr, col_global, val, y, offset, weight, has_intercept,
                                  binary_labels=(model_type == LOGISTIC_REGRESSION), dummy=dummy)
    D = 1 if dummy else int(num_features)   # the dummy zero feature of an intercept-only model occupies global index 0
    if not dummy and batch.col_global.size and (batch.col_global.min() < 0 or batch.col_global.max() >= D):
        raise ValueError(f"feature index outside [0, {D})")
    # whether the variances can be computed is decided before any training (a job that trains for its whole budget and
    # then dies on the variances loses the model)
    if variance_mode is not None:
        check_variance_request(str(variance_mode).upper(), D + (1 if has_intercept else 0), _world_size(group))
    s = self.solver
    packed = s.pack(batch, has_intercept=has_intercept)
    opts = SolverOptions(l2=l2, regularize_bias=bool(regularize_bias) and bool(has_intercept), has_intercept=has_intercept, m=m,
                         max_iter=max_iter, ftol=tolerance, threshold=0.0, sum_loss=True, linear=(model_type == LINEAR_REGRESSION))
    ic = 1 if has_intercept else 0
    t0 = None
    if theta0 is not None:
        full = np.zeros(D + ic)
        th = np.asarray(theta0, np.float64)
        if dummy:
            full[D:] = th[-ic:] if ic else []
        else:
            full[:] = th
        t0 = s.torch.from_numpy(full).to(s.device)
    prob = _SteppingProblem(s, packed, D, opts, t0)
    all_reduce = None
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if dist.get_backend(group) == "nccl":          # RCCL, in place on the device buffer, ordered on the stream
                def all_reduce(t):
                    dist.all_reduce(t, group=group)
                all_reduce.device_ordered = True
            else:                                          # e.g. gloo: staged through the host
                def all_reduce(t):
                    h = t.cpu()
                    dist.all_reduce(h, group=group)
                    t.copy_(h)
    except ImportError:
        pass
    status = run_stepping_loop(prob, all_reduce)
    theta, info = prob.result()
    info["status"] = status
    if variance_mode is not None:
        # the variances of the thresholded model on the training data, as the reference's scoring pass after training computes
        # them (fixed_effect_lr_lbfgs_model.py:648-661, 271-305, 451-463)
        th = np.where(np.abs(theta) <= threshold, 0.0, theta)
        info["variances"] = _variances(s, prob, batch, th, D, has_intercept, float(l2), bool(regularize_bias) and bool(has_intercept),
                                       str(variance_mode).upper(), all_reduce, group, packed=packed, dummy=dummy)
        if dummy:
            info["variances"] = info["variances"][D:]
    if dummy:
        theta = theta[D:]
    if return_problem:
        return theta, info, prob
    prob.close()
    return theta, info


# FULL densifies a (D + 1) x (D + 1) Hessian, as the reference does (fixed_effect_lr_lbfgs_model.py:291, 457: no limit there but
# memory). Up to FULL_VARIANCE_HOST_MAX coefficients the matrix is built and inverted on the host exactly as the reference does
# (several workers: summed with an all-reduce first); above that and up to FULL_VARIANCE_DEVICE_MAX on the device (tiled Cholesky
# and inverse over the whole MI355X, csrc/re_variance_big.hip). Several workers (round 4): every worker builds the dense curvature
# matrix of its shard on its device (gdmix_fe_hessian_dense), scatters it into the common index space, one all-reduce of the
# P x P matrix (RCCL), and every worker factors the sum (gdmix_fe_variance_of_hessian) — replicated, as scipy is in the reference.
FULL_VARIANCE_HOST_MAX = 4096
FULL_VARIANCE_DEVICE_MAX = 16384
FULL_VARIANCE_MAX_FEATURES = FULL_VARIANCE_HOST_MAX   # (the name round 2 used)


def _world_size(group=None):
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(group)
    except ImportError:
        pass
    return 1


def check_variance_request(mode, P, num_workers):
    """Raise before any training if the variances asked for cannot be computed for a model of P coefficients."""
    if mode == "SIMPLE":
        return
    if mode != "FULL":
        raise ValueError(f"unknown variance mode {mode!r}")
    if P > FULL_VARIANCE_DEVICE_MAX:
        raise ValueError(f"fixed_effect_variance_mode FULL inverts a dense {P} x {P} matrix; at most {FULL_VARIANCE_DEVICE_MAX} "
                         "coefficients (use SIMPLE for larger models)")


def _variances(solver, prob, batch, theta, D, has_intercept, l2, regularize_bias, mode, all_reduce, group, packed=None, dummy=False):
    """variance of every coefficient (intercept last). SIMPLE: 1 / (diag(X~' D X~) + l2 [regularised] + 1e-12), the diagonal by two
    more streaming passes on the device and the same all-reduce as an evaluation. FULL: diag((X~' D X~ + (l2 + 1e-12) I - l2
    [intercept unregularised])^-1): the dense matrix is built on the host from the shard (scipy), summed over the workers and
    inverted with numpy, exactly as the reference does (:296-305, 457-463) — only sensible for small feature spaces."""
    eps = 1.0e-12
    ic = 1 if has_intercept else 0
    P = D + ic
    if mode == "SIMPLE":
        t = solver.torch
        prob.hessian_diag(t.from_numpy(np.ascontiguousarray(theta, np.float64)).to(solver.device))
        buf = prob.reduce_tensor()
        if all_reduce is not None:
            all_reduce(buf)
        H = buf[:P].cpu().numpy().copy()
        H += l2
        if ic and not regularize_bias:
            H[-1] -= l2
        return 1.0 / (H + eps)
    if mode != "FULL":
        raise ValueError(f"unknown variance mode {mode!r}")
    check_variance_request(mode, P, _world_size(group))
    if P > FULL_VARIANCE_HOST_MAX and _world_size(group) > 1:
        return _full_variances_several_workers(solver, packed, theta, D, has_intercept, l2, regularize_bias, group, dummy)
    if P > FULL_VARIANCE_HOST_MAX:
        # on the device, in the shard's local index space (intercept first): the random-effect FULL variance of a one-entity
        # batch is this very matrix (binary_logistic_regression.py:181-187 = fixed_effect_lr_lbfgs_model.py:296-305, 457-463).
        # Features without a non-zero in the shard have a zero row and column in X~' D X~: their variance is 1 / (l2 + 1e-12).
        uniq = packed.unique_global().cpu().numpy()
        local = to_local(theta, uniq, D, has_intercept, dummy)
        o = SolverOptions(l2=l2, regularize_bias=regularize_bias, has_intercept=has_intercept)
        v_local = solver.variance_full(packed, o, local).cpu().numpy()
        out = np.full(P, 1.0 / (l2 + eps))
        out[uniq] = v_local[ic:]
        if ic:
            out[D] = v_local[0]
        return out
    import scipy.sparse as sp
    n = batch.N
    X = sp.csr_matrix((batch.val.astype(np.float64), batch.col_global, batch.row_nnz_ptr), shape=(n, max(D, 1)))[:, :D]
    if ic:
        X = sp.hstack([X, sp.csr_matrix(np.ones((n, 1)))], format="csr")
    z = X @ theta + batch.offset.astype(np.float64)
    rho = 1.0 / (1.0 + np.exp(-z))
    d = rho * (1.0 - rho) * (1.0 if batch.weight is None else batch.weight.astype(np.float64))
    H = np.asarray((X.T @ X.multiply(d[:, None])).todense(), np.float64)
    try:
        import torch
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            ht = torch.from_numpy(H)
            if dist.get_backend(group) == "nccl":
                hd = ht.to(solver.device)
                dist.all_reduce(hd, group=group)
                H = hd.cpu().numpy()
            else:
                dist.all_reduce(ht, group=group)
                H = ht.numpy()
    except ImportError:
        pass
    H = H + np.diag([l2 + eps] * P)
    if ic and not regularize_bias:
        H[-1, -1] -= l2
    return np.diagonal(np.linalg.inv(H)).copy()


def _full_variances_several_workers(solver, packed, theta, D, has_intercept, l2, regularize_bias, group, dummy=False):
    """FULL variances on the device with the Hessian summed over the workers (fixed_effect_lr_lbfgs_model.py:291-305, 384-389,
    457-463). Coefficient order: features 0 .. D-1, intercept last."""
    import torch
    import torch.distributed as dist
    ic = 1 if has_intercept else 0
    P = D + ic
    uniq_dev = packed.unique_global()
    uniq = uniq_dev.cpu().numpy()
    p_l = uniq.size + ic
    ld = (P + 63) // 64 * 64
    # the summed matrix and the factorisation's work area are ld x ld doubles each, the local matrix another one at most: refuse with a
    # clear message rather than die in an allocation after the whole training
    need = 3 * ld * ld * 8 + (1 << 28)
    if solver.device.type == "cuda":
        # free = what the driver reports + what torch's caching allocator holds without using it (after a whole training run that can
        # be most of the HBM: ADVICE r5 — the driver's figure alone refused allocations that would have succeeded)
        free = torch.cuda.mem_get_info(solver.device)[0] + torch.cuda.memory_reserved(solver.device) - torch.cuda.memory_allocated(solver.device)
        if free < need:
            raise MemoryError(f"fixed_effect_variance_mode=FULL with {P} coefficients needs {need / 1e9:.1f} GB of free device memory "
                              f"({free / 1e9:.1f} GB free)")
    Hg = torch.zeros((ld, ld), dtype=torch.float64, device=solver.device)
    if not dummy and p_l > 0:     # (a worker without data trains on one weight-0 sample: its curvature is exactly zero — nothing to build or scatter)
        local = to_local(theta, uniq, D, has_intercept, dummy)
        Hl = solver.hessian_dense(packed, local, has_intercept)            # [ld_l, ld_l], local order: intercept first
        idx = torch.cat([torch.full((ic,), D, dtype=torch.int64, device=solver.device), uniq_dev.to(torch.int64)])   # local -> global coefficient
        Hg.index_put_((idx[:, None], idx[None, :]), Hl[:p_l, :p_l])       # one 2-D scatter into the common index space
        del Hl
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(Hg, group=group)
    else:      # (gloo: the two-worker tests on one device)
        h = Hg.cpu()
        dist.all_reduce(h, group=group)
        Hg = h.to(solver.device)
    unreg = D if (ic and not regularize_bias) else -1
    return solver.variance_of_hessian(Hg, P, l2, unreg).cpu().numpy()


FixedEffectDeviceSolver.fit_stepping = _fit_stepping
