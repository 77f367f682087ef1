/*
 * gdmix_re.h — C ABI of libgdmix_re.so, the MI355X-native random-effect (RE) trainer hot path.
 *
 * The reference (linkedin/gdmix, Python/TF/scipy, CPU only) has no FFI; the boundary this library
 * replaces is the body of three Python call sites (paths relative to the reference checkout):
 *
 *   prepare_jobs            gdmix-trainer/src/gdmix/models/custom/scipy/job_consumers.py:161-296
 *        -> gdmix_re_pack       (per-entity np.unique / local indexing / COO build, :243-250)
 *   BinaryLogisticRegressionTrainer.fit
 *                           gdmix-trainer/src/gdmix/models/custom/binary_logistic_regression.py:191-239
 *        -> gdmix_re_solve      (_loss :84-110, _gradient :121-131, scipy fmin_l_bfgs_b :223-231,
 *                                _compute_variance :144-189, threshold model_utils.py:4-12)
 *   BinaryLogisticRegressionTrainer.predict_proba(return_logits=True)
 *                           binary_logistic_regression.py:241-262, job_consumers.py:138-152
 *        -> gdmix_re_score
 *   Math.abs(id.toString.hashCode) % numPartitions
 *                           gdmix-data/src/main/scala/com/linkedin/gdmix/utils/PartitionUtils.scala:31-37
 *        -> gdmix_java_partition_id / gdmix_java_string_hash
 *
 * Conventions
 *   - every function returns 0 on success or a negative GDMIX_RE_E* code; nothing throws or aborts;
 *     gdmix_re_last_error() returns a thread-local NUL-terminated message for the last failure.
 *   - the caller owns every buffer. Device buffers may come from any allocator bound to the context's
 *     HIP device (the Python host passes torch tensor data_ptr()s). The library never frees or
 *     retains caller memory after the work enqueued on `stream` has completed.
 *   - all device work is enqueued on the caller's `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and is asynchronous with respect to the host unless stated otherwise.
 *   - a context is bound to one HIP device; calls on one context must be serialised by the caller.
 *   - all arithmetic of the solver is IEEE fp64 on fp32-valued inputs, as in the reference.
 */
#ifndef GDMIX_RE_H_
#define GDMIX_RE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDMIX_RE_ABI_VERSION 12

#if defined(__GNUC__)
#define GDMIX_API __attribute__((visibility("default")))
#else
#define GDMIX_API
#endif

/* error codes */
#define GDMIX_RE_OK          0
#define GDMIX_RE_EINVAL    (-1)   /* bad argument */
#define GDMIX_RE_EHIP      (-2)   /* a HIP runtime call failed (no device, launch failure, ...) */
#define GDMIX_RE_ENOMEM    (-3)   /* workspace too small */
#define GDMIX_RE_ERANGE    (-4)   /* an entity exceeds an int32 per-entity limit */

/* per-entity solver status, mirrors scipy fmin_l_bfgs_b's task/warnflag */
#define GDMIX_RE_ST_PGTOL     0   /* CONVERGENCE: NORM OF PROJECTED GRADIENT <= PGTOL      */
#define GDMIX_RE_ST_FACTR     1   /* CONVERGENCE: REL_REDUCTION_OF_F <= FACTR*EPSMCH         */
#define GDMIX_RE_ST_MAXITER   2   /* STOP: TOTAL NO. of ITERATIONS REACHED LIMIT             */
#define GDMIX_RE_ST_MAXFUN    3   /* STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT    */
#define GDMIX_RE_ST_ABNORMAL  4   /* ABNORMAL_TERMINATION_IN_LNSRCH                          */
#define GDMIX_RE_ST_ABORTED   9   /* device-wide kernel gave up waiting at a barrier (never expected; the result is invalid) */
#define GDMIX_RE_ST_ABORTED_PEER 10 /* fixed effect with several workers: ANOTHER worker's step was aborted; its mark came with the
                                     * all-reduce and every worker stops in the same evaluation (the result is invalid) */
/* "converged" for the entities/sec metric = status in {PGTOL, FACTR, MAXITER}: the reference treats
 * all three as a finished model (job_consumers.py:36-63 never looks at warnflag). */

#define GDMIX_RE_VAR_NONE    0
#define GDMIX_RE_VAR_SIMPLE  1    /* 1/(diag(X'DX) + l2 + 1e-12), binary_logistic_regression.py:175-180 */
#define GDMIX_RE_VAR_FULL    2    /* diag(inv(X'DX + (l2+1e-12)I)),                         :181-187 */

typedef struct gdmix_re_ctx gdmix_re_ctx;

/* ---- raw entity-grouped batch: exactly what prepare_jobs slices per entity ------------------------
 * E entities, N samples, Z non-zeros, entity-major then sample-major (the order of the TF sparse
 * tensors in job_consumers.py:176-199). Pointers are DEVICE pointers for gdmix_re_pack and HOST
 * pointers for the oracle. */
typedef struct {
  int64_t E, N, Z;
  const int64_t* ent_row_ptr;   /* [E+1] sample offsets of each entity                              */
  const int64_t* row_nnz_ptr;   /* [N+1] non-zero offsets of each sample                            */
  const int64_t* col_global;    /* [Z]   global feature index ("<bag>_indices")                     */
  const float*   val;           /* [Z]   feature value        ("<bag>_values")                      */
  const float*   y;             /* [N]   label, exactly 0.0f or 1.0f (fit() asserts this, :208)     */
  const float*   offset;        /* [N]   fixed-effect score (offset_column_name)                    */
  const float*   weight;        /* [N]   sample weight, or NULL => ones (job_consumers.py:255-256)  */
} gdmix_re_raw_batch;

/* ---- the same batch in its 32-bit hand-over form (what crosses PCIe) -------------------------------
 * Counts instead of pointers, int32 or uint16 feature ids (gdmix_re_pack requires them below 2^31 anyway), byte
 * labels: 0.47 of the bytes of the raw form for C2. gdmix_re_widen rebuilds the raw arrays in HBM; val / offset / weight
 * are used in place. The counts must add up: sum(ent_n) == N, sum(row_nnz) == Z. DEVICE pointers. */
typedef struct {
  int64_t E, N, Z;
  const int32_t* ent_n;         /* [E] samples of each entity                                          */
  const void*    row_nnz;       /* [N] non-zeros of each sample, unsigned, row_nnz_width bytes each     */
  int32_t        row_nnz_width; /* 1, 2 or 4                                                            */
  int32_t        y_width;       /* 1: y is uint8 0/1 (logistic labels); 4: y is float                   */
  int32_t        col_width;     /* 4: col_global is int32; 2: uint16 (feature spaces of at most 65536)  */
  int32_t        reserved;
  const void*    col_global;    /* [Z]                                                                  */
  const float*   val;           /* [Z]                                                                  */
  const void*    y;             /* [N]                                                                  */
  const float*   offset;        /* [N]                                                                  */
  const float*   weight;        /* [N] or NULL                                                          */
} gdmix_re_wire_batch;

/* ---- packed ragged CSR(+CSC) batch in HBM, produced by gdmix_re_pack ------------------------------
 * All pointers point into the caller-provided workspace (or alias the raw batch for y/offset/weight).
 * Entity e owns
 *   samples       [ent_row_ptr[e],  ent_row_ptr[e+1])            n_e
 *   non-zeros     [ent_nnz_ptr[e],  ent_nnz_ptr[e+1])            z_e  (must be < 2^31)
 *   features      [ent_feat_ptr[e], ent_feat_ptr[e+1])           d_e  distinct global indices
 *   coefficients  [ent_feat_ptr[e] + e*has_intercept, +p_e)      p_e = d_e + has_intercept
 *   row_ptr       [ent_row_ptr[e] + e,  + n_e + 1)   entity-relative nnz offsets (CSR)
 *   col_ptr       [ent_nnz_ptr[e] + e,  + d_e + 1)   entity-relative nnz offsets (CSC); addressed by the
 *                 entity's non-zero offset (d_e <= z_e) so that packing does not wait for the prefix sum of d_e
 * unique_global is sorted ascending inside an entity (np.unique, job_consumers.py:243); csr_col are
 * local indices 0..d_e-1; the CSC copy is sorted by (local col, sample) so that the transposed
 * product X'r is an ordered, atomic-free per-coefficient sum. */
typedef struct {
  int64_t E, N, Z, D;           /* D = sum of d_e; valid on the host once the pack stream is synced */
  const int64_t* ent_row_ptr;   /* [E+1] */
  int64_t*       ent_nnz_ptr;   /* [E+1] */
  int64_t*       ent_feat_ptr;  /* [E+1] */
  int32_t*       row_ptr;       /* [N+E] */
  int32_t*       csr_col;       /* [Z]   */
  float*         csr_val;       /* [Z]   */
  int32_t*       col_ptr;       /* [Z+E] sparse: d_e + 1 entries at ent_nnz_ptr[e] + e */
  int32_t*       csc_row;       /* [Z]   entity-relative sample index */
  float*         csc_val;       /* [Z]   */
  int32_t*       unique_global; /* [D]   (allocated Z) local -> global feature index; int32 since ABI 12 (was int64: global feature
                                 *       indices are below 2^31 on this path, and the array is written, copied to the host and read there
                                 *       once per partition — half the bytes each time) */
  const float*   y;             /* [N]   */
  const float*   offset;        /* [N]   */
  const float*   weight;        /* [N] or NULL */
  /* solver-owned scratch inside the workspace (written by gdmix_re_solve) */
  int32_t*       order;         /* [E]   entity ids grouped by size class (solver launch order)     */
  int32_t*       cls_tmp;       /* [E]   size class of each entity                                  */
  int32_t*       class_count;   /* [6*NUM_CLASSES] per-class counts / bases / cursors / largest nnz / total nnz (u64) (device) */
  void*          scratch;       /* pack-time sort scratch, free for reuse once pack has returned    */
  size_t         scratch_bytes;
  int32_t        max_p, max_n, max_nnz;  /* per-entity maxima over the batch (host, after pack)     */
} gdmix_re_packed;

#define GDMIX_RE_NUM_CLASSES 40

/* ---- solver options (defaults = REParams/LRParams defaults + scipy defaults) ----------------------
 * base_lr_params.py:22-27, binary_logistic_regression.py:223-231 (pgtol/maxfun/maxls are scipy's). */
typedef struct {
  double  l2;               /* l2_reg_weight                          default 1.0   */
  int32_t regularize_bias;  /*                                        default 1     */
  int32_t has_intercept;    /*                                        default 1     */
  int32_t m;                /* num_of_lbfgs_curvature_pairs           default 10    */
  int32_t max_iter;         /* num_of_lbfgs_iterations                default 100   */
  int32_t maxfun;           /* scipy default                          15000         */
  int32_t maxls;            /* scipy default                          20            */
  double  ftol;             /* = factr*eps = lbfgs_tolerance          default 1e-12 */
  double  pgtol;            /* scipy default                          1e-5          */
  int32_t variance_mode;    /* GDMIX_RE_VAR_*                         default NONE  */
  double  threshold;        /* sparsity_threshold applied to theta_thr, default 1e-4 (model_utils.py:4-12) */
  /* The two switches below turn the per-entity objective into the fixed-effect one
   * (fixed_effect_lr_lbfgs_model.py:309-392): a batch with one "entity" = one worker's shard. Defaults 0.
   * Non-default values route every entity to the team kernels. */
  int32_t sum_loss;         /* 1: f = sum_i w_i l_i + (l2/2)|theta_reg|^2, not divided by n (:363-381)            */
  int32_t linear;           /* 1: l_i = (y_i - z_i)^2 (linear regression, :356-358) instead of the logistic loss */
} gdmix_re_opts;

/* fills *o with the defaults above */
GDMIX_API void gdmix_re_default_opts(gdmix_re_opts* o);

/* ---- per-entity / per-coefficient outputs of a solve (device pointers; any may be NULL) ----------- */
typedef struct {
  double*  theta;      /* [P]  raw L-BFGS result (result[0] of fmin_l_bfgs_b), local index space     */
  double*  theta_thr;  /* [P]  after threshold_coefficients(|x|<=threshold -> 0)                     */
  double*  variance;   /* [P]  when opts.variance_mode != NONE                                       */
  double*  fval;       /* [E]  final objective                                                       */
  double*  gnorm;      /* [E]  max|g| at the returned point                                          */
  int32_t* nit;        /* [E]                                                                        */
  int32_t* nfev;       /* [E]                                                                        */
  int32_t* status;     /* [E]  GDMIX_RE_ST_*                                                         */
} gdmix_re_result;

GDMIX_API int  gdmix_re_abi_version(void);
/* (ABI 12) Hash (16 hex digits) of the sources this binary was compiled from — the .hip units, csrc/*.hpp and the two C headers;
 * gdmix_amd/build.py: source_id(). The library travels prebuilt next to its sources; the Python loader refuses a binary whose id is
 * not the hash of the sources beside it, and build.needs_build() compares this id, not modification times. */
GDMIX_API const char* gdmix_re_build_id(void);
GDMIX_API const char* gdmix_re_last_error(void);

/* (ABI 12) Two PROCESSES on one device — the reference starts num_of_consumers processes per worker
 * (random_effect_lr_lbfgs_model.py:103,214-217) and TF_CONFIG may list more workers than the box has GPUs. The solver's persistent grids
 * (team tiers, tall teams) need all their workgroups resident; two of them from two processes can starve each other. They are chained by a
 * file lock per device (<GDMIX_RE_LOCK_DIR or /tmp>/gdmix_re_grid_<pci bus id>.lock, held while a process has such a grid in flight; a
 * turnstile file keeps two processes alternating; GDMIX_RE_GRID_LOCK=0 turns it off). These three entry points are that lock for a named
 * key, host only — what tests/test_grid_lock.py drives from several processes. acquire blocks until this process may launch (it counts:
 * one release per acquire); stats returns how often the process took the file lock and how often a launch rode on a lock it already held. */
/* 1 if ANOTHER process has a context on this context's device right now (each process holds a record lock on a per-device file from its
 * first gdmix_re_create on; same directory and switch as above), 0 if not, < 0 on error. What gdmix_fe_create asks before it chooses the
 * one-launch step (whose workgroups wait for each other: next to another process's persistent grid it takes the three-launch form). */
GDMIX_API int gdmix_re_device_shared(gdmix_re_ctx* ctx);
GDMIX_API int gdmix_re_grid_lock_acquire(const char* key);
GDMIX_API int gdmix_re_grid_lock_release(const char* key);
GDMIX_API int gdmix_re_grid_lock_stats(const char* key, int64_t* takes, int64_t* rides);

GDMIX_API int  gdmix_re_create(int hip_device, gdmix_re_ctx** out);
GDMIX_API void gdmix_re_destroy(gdmix_re_ctx* ctx);

/* Bytes of device workspace gdmix_re_pack needs for a batch of this shape (upper bound; host-only). */
GDMIX_API size_t gdmix_re_pack_workspace_bytes(int64_t E, int64_t N, int64_t Z);

/* Pack: per entity, unique-sort the global feature indices, re-index the non-zeros locally, build the
 * CSR and CSC copies and the size-class launch order — all on the device. Fills *out (a host struct
 * of device pointers into `workspace`). Synchronises `stream` once to read back D and the class
 * counts. replaces job_consumers.py:209-258 (enable_local_indexing=True form; the reference's
 * global-indexing form yields identical coefficients on the entity's support). */
GDMIX_API int gdmix_re_pack(gdmix_re_ctx* ctx, const gdmix_re_raw_batch* raw_dev, int has_intercept,
                  void* workspace, size_t workspace_bytes, gdmix_re_packed* out, void* stream);

/* (ABI 11) The last kernel of a pack compacts every entity's unique feature ids into `unique_global` — the one output no solve
 * kernel reads (it names the coefficients for the model table, job_consumers.py:243, and maps a fixed-effect shard's columns). With
 * `enabled` != 0 gdmix_re_pack queues that kernel on a side stream of the context and returns without waiting for it, so it runs
 * NEXT to the gdmix_re_solve the caller queues behind the pack: a copy-shaped kernel beside kernels bound by their arithmetic.
 * Every other member of gdmix_re_packed is ordered on `stream` as before. `unique_global` is ordered on the stream of the next of
 * these calls on the same context: gdmix_re_solve (on return the stream is behind the compaction as it is behind the solve),
 * gdmix_re_score, gdmix_re_variance_full, gdmix_fe_create, gdmix_re_pack (the next batch), gdmix_re_pack_join. A caller that reads
 * unique_global itself, or frees / reuses the workspace, without one of them in between calls gdmix_re_pack_join(ctx, stream)
 * first. Default off (everything stream-ordered when gdmix_re_pack returns); gdmix_amd/solver.py switches it on. Results are the
 * same bits either way. A context without side streams ignores the request. */
GDMIX_API int gdmix_re_set_defer_unique(gdmix_re_ctx* ctx, int enabled);
GDMIX_API int gdmix_re_pack_join(gdmix_re_ctx* ctx, void* stream);

/* Wire form -> raw form on the device (two prefix sums and two widening copies, ~0.3 ms for C2), enqueued on
 * `stream`; fills *out (a host struct of device pointers into `workspace` and into the wire arrays). The result
 * feeds gdmix_re_pack on the same stream. */
GDMIX_API size_t gdmix_re_widen_workspace_bytes(int64_t E, int64_t N, int64_t Z);
GDMIX_API int gdmix_re_widen(gdmix_re_ctx* ctx, const gdmix_re_wire_batch* wire_dev, void* workspace, size_t workspace_bytes,
                   gdmix_re_raw_batch* out, void* stream);

/* Solve every entity of the packed batch: the whole L-BFGS loop runs on the device, one wavefront
 * (or workgroup, for entities that do not fit a wavefront's LDS budget) per entity.
 * theta0: [P] warm-start coefficients in local index space, or NULL => zeros (fit():220-221). */
GDMIX_API int gdmix_re_solve(gdmix_re_ctx* ctx, const gdmix_re_packed* batch, const gdmix_re_opts* opts,
                   const double* theta0, const gdmix_re_result* out, void* stream);

/* Bytes of device scratch gdmix_re_solve needs beyond the pack workspace (0 if none). */
GDMIX_API size_t gdmix_re_solve_scratch_bytes(const gdmix_re_packed* batch, const gdmix_re_opts* opts);
GDMIX_API int    gdmix_re_set_scratch(gdmix_re_ctx* ctx, void* scratch, size_t bytes);

/* variance_mode FULL on its own: diag((X~' D X~ + (l2 + 1e-12) I - l2 e0 e0' [intercept unregularised])^-1) of every entity of the
 * batch at `theta` ([P], local index space), D = rho (1 - rho) w (binary_logistic_regression.py:181-187) -> variance [P].
 * Entities up to 16 384 coefficients (a dense p x p matrix per entity, as the reference builds). Needs the scratch of
 * gdmix_re_solve_scratch_bytes with variance_mode FULL. Also what the fixed-effect stage uses for its FULL variances on one
 * worker (fixed_effect_lr_lbfgs_model.py:296-305, 457-463: the same matrix, intercept last instead of first). */
GDMIX_API int gdmix_re_variance_full(gdmix_re_ctx* ctx, const gdmix_re_packed* batch, const gdmix_re_opts* opts, const double* theta,
                                     double* variance, void* stream);

/* Score: logit[i] = x_i . theta_e + offset[i] (fp64 accumulate, stored fp32 as the score Avro
 * does, io_utils.py:367-375), logit_per_coord[i] = logit[i] - offset[i] (job_consumers.py:145-150).
 * has_model: [E] uint8, 0 => entity has no model and logit = offset (job_consumers.py:145-146);
 * NULL => all entities have one. theta is in the batch's local index space [P]. */
GDMIX_API int gdmix_re_score(gdmix_re_ctx* ctx, const gdmix_re_packed* batch, int has_intercept,
                   const double* theta, const uint8_t* has_model,
                   float* logit, float* logit_per_coord, void* stream);

/* Tuning/testing knob: entities whose LDS footprint exceeds `bytes` are solved by the
 * workgroup-per-entity kernel (0 => every entity). Default and maximum 65536. */
GDMIX_API int gdmix_re_set_wave_lds_limit(gdmix_re_ctx* ctx, int bytes);

/* Tuning/testing knob: which per-entity kernels the solver may use. bit 1 = LDS-resident wavefront kernel (any m), bit 2 = the
 * group kernels (several entities per wavefront); the tall kernel and the team kernels are always available. Default 7. Bit 0 was
 * round 1's register-resident one-entity-per-wavefront kernel: unreachable under default routing once the group kernels covered
 * p <= 2048 (tests/test_gpu_parity.py::test_default_routing_reaches_only_these_classes), removed in round 4 with its thirteen size
 * classes (ABI 8, GDMIX_RE_NUM_CLASSES 38; 39 since the tall team class of ABI 9); the bit is accepted and ignored. (Round 2's bit 3, team kernels with the history in
 * registers, went in round 3 - profiles/r03_zipf_register_team_kernels.txt.) */
GDMIX_API int gdmix_re_set_kernel_mask(gdmix_re_ctx* ctx, int mask);

/* The head of a Zipf-distributed partition is solved by a persistent kernel (one workgroup per CU) split into teams
 * of CUs, every team taking the next entity of its tier as it becomes free, largest first. Tiers by non-zeros:
 *   [team_nnz, 8 team_nnz)        128 teams of 2 CUs
 *   [8 team_nnz, 128 team_nnz)     32 teams of 8 CUs
 *   [128 team_nnz, giant_nnz)       8 teams of 32 CUs
 *   >= giant_nnz                    the whole device, one entity after another
 * (these entities are bound by synchronisation and memory latency, not by throughput: small teams, many at a time).
 * Smaller entities: one workgroup each. Defaults team_nnz 16384, giant_nnz 16777216; 0 disables the tiers above /
 * the device-wide tier. Results do not depend on the thresholds beyond summation order. */
GDMIX_API int gdmix_re_set_giant_nnz(gdmix_re_ctx* ctx, int64_t giant_nnz);
GDMIX_API int gdmix_re_set_team_nnz(gdmix_re_ctx* ctx, int64_t team_nnz);

/* Tall and skinny entities — at most 64 coefficients and at least `min_n` samples (MovieLens per-user / per-movie random
 * effects: up to ~54 k samples for 25 coefficients) — are solved by one workgroup each, the samples over all its lanes, the
 * L-BFGS driver replicated in every wavefront's registers (csrc/re_solve_tall.hip). 0 = never. Results do not depend on the
 * threshold beyond summation order. */
#define GDMIX_RE_TALL_MIN_N_DEFAULT 32
GDMIX_API int gdmix_re_set_tall_min_n(gdmix_re_ctx* ctx, int min_n);
/* Tall entities of at least `split_n` samples get a CU each (a workgroup of eight wavefronts); smaller ones share a CU,
 * eight single-wavefront workgroups at a time, so that one entity's L-BFGS driver (a latency-bound chain in one
 * wavefront) runs while the others' passes over their samples do. Until this is called the split is the default, lowered per
 * batch to 2 048 / 1 024 / 512 when the eight-wavefront class stays small that way; a split set by the caller is kept as it is,
 * also when it equals the default (the way to switch the per-batch choice off); split_n = 0 returns to the adaptive default. */
#define GDMIX_RE_TALL_SPLIT_N_DEFAULT 4096
GDMIX_API int gdmix_re_set_tall_split_n(gdmix_re_ctx* ctx, int split_n);
/* The tallest entities of a batch get a TEAM of four workgroups (four CUs of one XCD) that share the pass over one entity's
 * samples: a share of a strongly scaled MovieLens job lasts as long as ONE workgroup needs for its most rated title
 * (ABI 9, GDMIX_RE_NUM_CLASSES 39). `team_n` > 0: eight-wavefront tall entities of at least team_n, 2 team_n or 4 team_n samples
 * - the lowest of the three that keeps the class within one round of teams on the device (a quarter of its CUs' worth of entities); a batch with
 * more entities than that above 4 team_n has no team class (it is bound by throughput, not by one entity's chain).
 * `team_n` < 0: every tall entity of at least -team_n samples (at least 64), no limit. 0 = never. The split of an entity's samples
 * over the four workgroups depends on its size alone, and a team's result agrees with the one-workgroup kernel's to rounding
 * (another summation order) — but WHICH of the two kernels an entity gets depends on the batch when team_n > 0 (and likewise
 * for the per-batch split below gdmix_re_set_tall_split_n's default): the same entity can come out with other last bits in
 * another batch. A caller that needs results independent of the batching (entity re-balancing, comparisons across partitionings)
 * pins both: gdmix_re_set_tall_split_n(ctx, GDMIX_RE_TALL_SPLIT_N_DEFAULT) and gdmix_re_set_tall_team_n(ctx,
 * -GDMIX_RE_TALL_TEAM_N_DEFAULT) (gdmix_amd/solver.py: pin_routing). A device with fewer than 32 CUs never gets the class. */
#define GDMIX_RE_TALL_TEAM_N_DEFAULT 8192
GDMIX_API int gdmix_re_set_tall_team_n(gdmix_re_ctx* ctx, int team_n);
/* (ABI 12, GDMIX_RE_NUM_CLASSES 40) The MID tall class: in a SMALL batch (a share of a strongly scaled job) the largest one-wavefront tall
 * entities below the split get a workgroup of four wavefronts with half a CU's LDS (they stay resident there; on one wavefront they are
 * streamed, and the share lasts as long as that one wavefront's chain: random_effect_driver.py:60-68 splits the partitions over the
 * workers, BASELINE config 3). OFF by default: measured, it makes those shares slower (the one-wavefront launch is bound by the
 * throughput of its many small entities, not by its longest chain, and a mid workgroup takes LDS away from four of them:
 * profiles/r06_ml20m_mid.txt); kept, parity-tested, for devices where that balance differs. mid_n < 0: chosen per batch on the device — the lowest of 256 / 384 / 512 / 768 / 1 024 / 1 536
 * samples that keeps the class within one round of its launch (two workgroups per CU), only when the one-wavefront class is small and the
 * split is not pinned; mid_n > 0: every one-wavefront tall entity of at least mid_n samples, whatever the batch (tests); 0: never. Same
 * caveat as the teams: the kernel an entity gets depends on the batch, its sums are added in another order (agreement to rounding);
 * gdmix_re_set_tall_split_n(ctx, default) pins this choice too. GDMIX_RE_TALL_MID=1 in the environment switches the per-batch class on. */
GDMIX_API int gdmix_re_set_tall_mid_n(gdmix_re_ctx* ctx, int mid_n);

/* Launch schedule of a solve. Size classes too small to fill the device always run next to the others on the context's side streams
 * (three, created with the context). `queues` > 1 (default 4 = the caller's stream + the three side streams; GDMIX_RE_SPREAD in the
 * environment sets the default of new contexts): the LARGE classes are dealt over that many streams in launch order as well, so that the
 * tail of one class launch (the entities with the most iterations) overlaps with the next class instead of idling the device; every
 * side stream is joined back into the caller's stream before gdmix_re_solve returns. 0 or 1: large classes one after another on the
 * caller's stream. A schedule changes the time, never a bit of the result; per-class durations (gdmix_re_last_solve_ms) of overlapped
 * launches stretch each other. */
GDMIX_API int gdmix_re_set_spread(gdmix_re_ctx* ctx, int queues);

/* Optional kernel timing: when enabled, gdmix_re_solve brackets each size class's kernel launch with
 * HIP events on the caller's stream; gdmix_re_last_solve_ms waits for them and returns the elapsed
 * milliseconds per class ([GDMIX_RE_NUM_CLASSES] floats, 0 for classes that were not launched). */
GDMIX_API int gdmix_re_set_timing(gdmix_re_ctx* ctx, int enabled);
GDMIX_API int gdmix_re_last_solve_ms(gdmix_re_ctx* ctx, float* ms_out);

/* Name of the kernel variant that solved size class c (for profiling reports), or NULL. */
GDMIX_API const char* gdmix_re_class_kernel_name(int c);

/* ---- B4: the upstream Spark partitioner's hash, bit-exact (host functions) ------------------------
 * hashCode over UTF-16 code units in wrapping int32; Math.abs(Int.MinValue) stays negative; Scala %
 * keeps the dividend's sign (PartitionUtils.scala:31-37). */
GDMIX_API int32_t gdmix_java_string_hash(const uint16_t* utf16, int64_t len);
GDMIX_API int32_t gdmix_java_partition_id(const uint16_t* utf16, int64_t len, int32_t num_partitions);
/* Batched device form for decimal int64 entity ids (id.toString of a Long): out[i] = partition id. */
GDMIX_API int gdmix_java_partition_ids_i64(gdmix_re_ctx* ctx, const int64_t* ids_dev, int64_t count,
                                 int32_t num_partitions, int32_t* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDMIX_RE_H_ */
