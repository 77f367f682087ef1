/* gdmix_fe.h — C ABI of the fixed-effect trainer in libgdmix_re.so.
 *
 * Replaces the body of FixedEffectLRModelLBFGS's training step
 *   _train_model_fn / _compute_loss_and_gradients / fmin_l_bfgs_b call
 *   gdmix-trainer/src/gdmix/models/custom/fixed_effect_lr_lbfgs_model.py:309-392, 394-430, 635-643
 * for one worker's shard resident in HBM:
 *
 *   f(theta) = sum_i w_i loss(y_i, x_i.w + b + offset_i) + (l2/2) |theta_reg|^2     theta = [w (num_features), b]
 *
 * (intercept LAST, :340-342; not divided by n; theta_reg excludes b unless regularize_bias, :367-369). The
 * reference evaluates value and gradient with TensorFlow on every worker, all-reduces both across workers
 * (two collectives, keys 0/1, each worker adding l2 |theta|^2 / (2 R), :375-381) and hands them to scipy's
 * fmin_l_bfgs_b, which runs replicated on every worker. Here one evaluation is
 *
 *   gdmix_fe_eval    local part of [gradient (num_features + 1), value] into one device buffer
 *   <caller>         ONE all-reduce (sum) of that buffer across workers (RCCL through torch.distributed);
 *                    nothing to do for a single worker
 *   gdmix_fe_step    adds the regulariser once, advances L-BFGS (replicated: every worker computes the same
 *                    step from the same reduced buffer), returns the solver status (-1 = evaluate again)
 *
 * so the host loop is `do { eval; all_reduce; } while (step() < 0)`. All arithmetic fp64 on fp32 data. The shard is
 * a one-entity batch packed by gdmix_re_pack (CSR + CSC copy, local feature ids + unique_global map); the
 * L-BFGS vectors live in the global coefficient space, which is common to all workers.
 *
 * Same conventions as gdmix_re.h: 0 / negative return codes, gdmix_re_last_error(), no exceptions, caller-owned
 * input buffers (which must stay alive until gdmix_fe_destroy), one context per device, calls serialised.
 */
#ifndef GDMIX_FE_H
#define GDMIX_FE_H

#include "gdmix_re.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gdmix_fe_problem gdmix_fe_problem;

/* shard: a packed batch with E == 1 (gdmix_re_pack; has_intercept of the pack must equal opts->has_intercept).
 * num_features: size of the global feature space D; coefficients are [D + has_intercept], intercept last.
 * opts: l2, regularize_bias, has_intercept, m (<= 10), max_iter, maxfun, maxls, ftol, pgtol, linear are used.
 * theta0: device pointer [D + has_intercept] or NULL (zeros).
 * The problem builds its own two copies of the non-zeros (8 B per non-zero each, 10 B when a unit of a pass spans more than 2^21
 * elements; temporary: 40 B per non-zero) and keeps reading the shard's y / offset / weight / unique_global, which must outlive it.
 * Synchronises the stream. Test hooks (environment): GDMIX_FE_CHUNK = entries per unit of a pass, GDMIX_FE_PACK=0 = the
 * three-array form of the entries, GDMIX_FE_COMPRESS = which passes may read units in the 6-byte form (bit 0 rows, bit 1 columns;
 * default 2). */
GDMIX_API int gdmix_fe_create(gdmix_re_ctx* ctx, const gdmix_re_packed* shard, int64_t num_features,
                              const gdmix_re_opts* opts, const double* theta0, gdmix_fe_problem** out, void* stream);
GDMIX_API void gdmix_fe_destroy(gdmix_fe_problem* p);

/* Local [gradient of the data term (D + has_intercept), value of the data term] at the current trial point. */
GDMIX_API int gdmix_fe_eval(gdmix_fe_problem* p, void* stream);

/* The buffer gdmix_fe_eval fills and gdmix_fe_step consumes: *count = D + has_intercept + 1 doubles. */
GDMIX_API double* gdmix_fe_reduce_buffer(gdmix_fe_problem* p, int64_t* count);

/* Diagonal of the data term's Hessian X~' D X~ of the shard at theta (device pointer [D + has_intercept], intercept last;
 * NULL = the current point), D_i = w_i rho_i (1 - rho_i), rho = sigmoid(x_i . w + b + offset_i) — what
 * fixed_effect_lr_lbfgs_model.py:271-296 accumulates batch by batch for fixed_effect_variance_mode = SIMPLE. Written to the
 * reduce buffer in place of the gradient (entries [0, D + has_intercept); the last entry is unused): the same all-reduce as
 * an evaluation makes it the whole data set's, and variance_j = 1 / (H_j + l2 [j regularised] + 1e-12) (:451-456). The
 * reference does this arithmetic in float32, this library in float64. Two more streaming passes over the shard. */
GDMIX_API int gdmix_fe_hessian_diag(gdmix_fe_problem* p, const double* theta, void* stream);

/* fixed_effect_variance_mode = FULL with several workers, on the device (round 4). The reference sums the workers' dense Hessians
 * and inverts the sum (fixed_effect_lr_lbfgs_model.py:291-305, 384-389, 457-463). Stage 1, per worker: the curvature part X~' D X~ of
 * the shard — `shard` is the one-entity packed batch gdmix_fe_create took — at theta_local ([d + has_intercept], the shard's LOCAL
 * order: intercept first, then its features in ascending global index, packed->unique_global) as a dense symmetric matrix H
 * [ld x ld] row-major, ld = d + has_intercept rounded up to a multiple of 64, no regulariser, zero on the padding. The caller
 * scatters H into the common (global) index space and all-reduces it (RCCL through torch.distributed). Stage 2: variance [p] =
 * diag((H + (l2 + 1e-12) I - l2 e_u e_u' [u = unregularised_index, -1: none])^-1) of the summed matrix H [ld x ld] (overwritten),
 * work = another ld x ld doubles. Tiled Cholesky + inverse on the whole device (csrc/re_variance_big.hip), p <= 16 384. */
GDMIX_API size_t gdmix_fe_hessian_dense_scratch_bytes(const gdmix_re_packed* shard);
GDMIX_API int gdmix_fe_hessian_dense(gdmix_re_ctx* ctx, const gdmix_re_packed* shard, int has_intercept, const double* theta_local,
                                     double* H, int64_t ld, void* scratch, size_t scratch_bytes, void* stream);
GDMIX_API int gdmix_fe_variance_of_hessian(gdmix_re_ctx* ctx, double* H, int64_t p, int64_t ld, double l2, int64_t unregularised_index,
                                           double* work, double* variance, void* stream);

/* Consumes the (all-reduced) buffer. *status: -1 = evaluate again, else GDMIX_RE_ST_*. Synchronises the stream. */
GDMIX_API int gdmix_fe_step(gdmix_fe_problem* p, void* stream, int32_t* status);

/* The same, stream-ordered (round 5): the step is enqueued, nothing is waited for; *seq (may be NULL) = the number of this step,
 * counted from 0 over the life of the problem. gdmix_fe_step_status(p, seq, &status) waits for THAT step only and returns its
 * status (the last 8 enqueued steps can be asked for). Once the driver has stopped, every later gdmix_fe_eval / step on the
 * problem is a no-op on the device (the kernels return on the stop flag) and reports the status of the stop, so a host loop may
 * run `lookahead` evaluations ahead of the status it has read and the device never idles between evaluations:
 *     for k = 0, 1, ...: eval; all_reduce; step_async -> k; if (k >= lookahead and step_status(k - lookahead) >= 0) break;
 * With several workers every worker takes the same decisions from the same reduced buffer, so all of them enqueue the same
 * number of all-reduces. gdmix_fe_solve is that loop for ONE worker (no all-reduce), inside the library: *status = the status
 * of the stop (-1: max_evals evaluations without one), *evals (may be NULL) = evaluations enqueued, no-ops included. */
GDMIX_API int gdmix_fe_step_async(gdmix_fe_problem* p, void* stream, int64_t* seq);
GDMIX_API int gdmix_fe_step_status(gdmix_fe_problem* p, int64_t seq, int32_t* status);
GDMIX_API int gdmix_fe_solve(gdmix_fe_problem* p, void* stream, int32_t lookahead, int64_t max_evals, int32_t* status, int64_t* evals);

/* Result after status >= 0: theta [D + has_intercept] (device pointer, may be NULL) and scalars (host, may be NULL). */
GDMIX_API int gdmix_fe_result(gdmix_fe_problem* p, double* theta, double* fval, double* gnorm, int32_t* nit,
                              int32_t* nfev, void* stream);

/* Scores of a raw shard under a global coefficient vector: replaces _predict / the scoring after training
 *   gdmix-trainer/src/gdmix/models/custom/fixed_effect_lr_lbfgs_model.py:214-306,406-440
 * score_i = x_i . w + b + offset_i, per_coord_i = score_i - offset_i, stored as float (the reference's Avro `float`). All
 * pointers are device pointers: row_nnz_ptr [n+1] / col_global / val are the sample-major arrays of the reader (NULL,
 * NULL, NULL for a model without a feature bag), offset [n] or NULL, theta [num_features + has_intercept] with the
 * intercept last. No pack is needed: the pass reads the shard once. Feature indices must lie in [0, num_features). */
GDMIX_API int gdmix_fe_score(gdmix_re_ctx* ctx, int64_t n, const int64_t* row_nnz_ptr, const int64_t* col_global, const float* val,
                             const float* offset, const double* theta, int64_t num_features, int has_intercept, float* score,
                             float* per_coord, void* stream);

/* Bytes of non-zero entries one row pass / one column pass of this problem reads (its own copies of the shard: 8 B per entry, 10 in the
 * three-array form; units in the 6-byte form of round 5 — values + 16-bit {key delta, accumulator} words — with their fillers and
 * padding). What the passes stream, next to the algorithmic 8 B per entry and pass the bench's roofline figure is quoted on. */
GDMIX_API int gdmix_fe_stream_bytes(gdmix_fe_problem* p, int64_t* rows_pass, int64_t* cols_pass);

/* Optional timing (HIP events on the launch stream): ms of the row pass (X theta) and the column pass (X'r) of the problem's SECOND
 * gdmix_fe_eval (its first, if there was only one) — not the last: with the status read a few steps late the last evaluations of a
 * solve are no-ops. */
GDMIX_API int gdmix_fe_last_eval_ms(gdmix_fe_problem* p, float* rows_ms, float* cols_ms);

#ifdef __cplusplus
}
#endif
#endif /* GDMIX_FE_H */
