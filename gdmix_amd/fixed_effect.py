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

Multi-worker training (the reference all-reduces value and gradient across workers, :375-381) is not built: one
MI355X holds 288 GB, i.e. shards the reference needs many CPU workers for.
"""
import numpy as np

from .batch import RawBatch
from .solver import REDeviceSolver, SolverOptions

LOGISTIC_REGRESSION = "logistic_regression"
LINEAR_REGRESSION = "linear_regression"


def shard_as_batch(row_nnz_ptr, col_global, val, y, offset=None, weight=None, has_intercept=True, binary_labels=True):
    """The shard as a one-entity RawBatch. An intercept-only model (no features at all) gets one dummy zero
    feature per sample, as the random-effect path does (job_consumers.py:213-218)."""
    row_nnz_ptr = np.asarray(row_nnz_ptr, np.int64)
    n = row_nnz_ptr.size - 1
    col = np.asarray(col_global, np.int64)
    v = np.asarray(val, np.float32)
    dummy = col.size == 0
    if dummy:
        if not has_intercept:
            raise ValueError("a model without features needs an intercept")
        row_nnz_ptr = np.arange(n + 1, dtype=np.int64)
        col = np.zeros(n, np.int64)
        v = np.zeros(n, np.float32)
    b = RawBatch(ent_row_ptr=np.array([0, n], np.int64), row_nnz_ptr=row_nnz_ptr, col_global=col, val=v,
                 y=np.asarray(y, np.float32), offset=np.zeros(n, np.float32) if offset is None else np.asarray(offset, np.float32),
                 weight=None if weight is None else np.asarray(weight, np.float32), uid=np.arange(n, dtype=np.int64),
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

    def fit(self, row_nnz_ptr, col_global, val, y, num_features, offset=None, weight=None, has_intercept=True, l2=1.0,
            regularize_bias=True, model_type=LOGISTIC_REGRESSION, theta0=None, max_iter=100, m=10, tolerance=1e-12):
        """-> (theta [num_features + has_intercept], intercept last; info dict with f, nit, nfev, status, gnorm)."""
        if model_type not in (LOGISTIC_REGRESSION, LINEAR_REGRESSION):
            raise ValueError(f"unknown model type {model_type!r}")
        batch, dummy = shard_as_batch(row_nnz_ptr, col_global, val, y, offset, weight, has_intercept,
                                      binary_labels=(model_type == LOGISTIC_REGRESSION))
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
        return theta, info
