// re_internal.hpp — host-side declarations shared by the translation units of libgdmix_re.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/gdmix_re.h"
#include "re_solve_core.hpp"
#include "re_solve_wreg.hpp"
#include "re_solve_quad.hpp"

namespace gdmix {

// Size classes: every entity is routed to the cheapest kernel variant that can hold it.
//   KIND_QUAD2/4    four entities per wavefront (one per 16-lane DPP row), p <= 32/64, state in registers
//   KIND_PAIR3/4    two entities per wavefront (one per pair of DPP rows), p <= 96/128
//   KIND_G64_3/4    one entity per wavefront, same register/LDS design, p <= 192/256
//   KIND_G128..512  one entity per workgroup of 2/4/8 wavefronts (cross-wave reduction stage), p <= 512/1024/2048
//   (KIND_WREG1/2/4/8: the register-resident one-entity-per-wavefront kernels of round 1; superseded by the group kernels, unreachable
//    under default routing since round 2 and removed in round 4 — the ids stay reserved)
//   KIND_WLDS       LDS-resident wavefront kernel (any p whose state fits 64 KiB of LDS, any m)
//   KIND_BLOCK      workgroup-per-entity kernel working out of a global scratch slot (anything)
//   KIND_TALL       workgroup-per-entity kernel for tall and skinny entities (p <= 64, n >= tall_min_n): samples over all
//                   lanes, the L-BFGS driver replicated in every wavefront's registers (re_solve_tall.hip)
// Each wavefront kind is split into LDS-footprint buckets so that small entities keep high occupancy.
enum { KIND_WREG1 = 0, KIND_WREG2 = 1, KIND_WREG4 = 2, KIND_WLDS = 3, KIND_BLOCK = 4, KIND_QUAD2 = 5, KIND_QUAD4 = 6, KIND_PAIR4 = 7, KIND_QUAD3 = 8, KIND_PAIR3 = 9, KIND_WREG8 = 10,
       KIND_G64_3 = 11, KIND_G64_4 = 12, KIND_G128_4 = 13, KIND_G256_4 = 14, KIND_G512_4 = 15, KIND_GRID = 16, KIND_G128_3 = 17, KIND_G256_3 = 18,
       KIND_TALL = 20, KIND_TALL_S = 21, KIND_TALL_L = 22, KIND_TALL_T = 23, KIND_TALL_M = 24 };   // (19: the register team kernels of round 2, removed in round 3)

// group kernels (several entities per wavefront): lanes per entity, coefficient slots per lane; 0 if not a group kind
__host__ __device__ inline int group_lanes(int kind) {
  return (kind == KIND_QUAD2 || kind == KIND_QUAD3 || kind == KIND_QUAD4) ? 16 : ((kind == KIND_PAIR3 || kind == KIND_PAIR4) ? 32 :
         ((kind == KIND_G64_3 || kind == KIND_G64_4) ? 64 : ((kind == KIND_G128_4 || kind == KIND_G128_3) ? 128 : ((kind == KIND_G256_4 || kind == KIND_G256_3) ? 256 : (kind == KIND_G512_4 ? 512 : 0)))));
}
__host__ __device__ inline int group_epl(int kind) {
  return kind == KIND_QUAD2 ? 2 : ((kind == KIND_QUAD3 || kind == KIND_PAIR3 || kind == KIND_G64_3 || kind == KIND_G128_3 || kind == KIND_G256_3) ? 3 : (group_lanes(kind) > 0 ? 4 : 0));
}
constexpr int GIANT_CLASS = GDMIX_RE_NUM_CLASSES - 1;   // device-wide kernel, one entity at a time
constexpr int TEAM8_CLASS = GDMIX_RE_NUM_CLASSES - 2;    // 8 teams of 32 CUs
constexpr int TEAM32_CLASS = GDMIX_RE_NUM_CLASSES - 3;   // 32 teams of 8 CUs
constexpr int TEAM128_CLASS = GDMIX_RE_NUM_CLASSES - 4;  // 128 teams of 2 CUs
constexpr int BLOCK_CLASS = GDMIX_RE_NUM_CLASSES - 5;
constexpr int TALL_CLASS = BLOCK_CLASS - 1;      // tall entities of at least tall_split_n samples: one workgroup of TALL_NW wavefronts per CU
constexpr int TALL_M_CLASS = BLOCK_CLASS - 2;    // (round 6) the largest of the smaller ones in a SMALL batch: workgroups of TALL_NW_MID wavefronts, two per CU
constexpr int TALL_S_CLASS = BLOCK_CLASS - 3;    // smaller ones: workgroups of TALL_NW_SMALL wavefronts, several per CU
constexpr int TALL_L_CLASS = BLOCK_CLASS - 4;    // ... and those that fit a twelfth of a CU's LDS: the lean variant, three wavefronts per SIMD
constexpr int TALL_T_CLASS = BLOCK_CLASS - 5;    // the tallest of a batch: TALL_TEAM_C workgroups (CUs of one XCD) share one entity's samples
constexpr int TALL_TEAM_C = 4;                   // workgroups per entity of that class (fixed: an entity's sums depend on the split)
constexpr int TALL_TEAM_MAX = 64;                // teams per launch
constexpr int TALL_TEAM_BYTES = 192 * 1024;      // device buffer of the context: the teams' exchange structures (re_solve_tall.hip)
constexpr int TALL_TEAM_MIN_N = 64;              // no team for fewer samples than this, whatever the caller asks for (every member gets samples)
#ifndef GDMIX_TALL_NW_SMALL
#define GDMIX_TALL_NW_SMALL 1
#endif
constexpr int TALL_NW = 8;
constexpr int TALL_NW_SMALL = GDMIX_TALL_NW_SMALL;
constexpr int TALL_NW_MID = 4;           // wavefronts of a mid workgroup,
constexpr int TALL_MID_WGS = 2;          // of which a CU holds two (half its LDS each: an entity of ~1 800 MovieLens samples stays resident)
constexpr int TALL_MAX_P = 64;      // coefficients (one per lane of the master wavefront)
constexpr int TALL_LEAN_WAVES = 3;                    // lean variant: wavefronts per SIMD,
constexpr int TALL_LEAN_WGS = 4 * TALL_LEAN_WAVES;    // one-wavefront workgroups per CU,
// accumulator sets per column and wavefront of the tall kernels: 16 in the one-wavefront workgroups (both variants the same: an
// entity's sums do not depend on which of them ran it), 32 for up to 32 coefficients in the eight-wavefront ones
__host__ __device__ constexpr int tall_sets(int nw, int p) { return (nw > 1 && p <= 32) ? 32 : 16; }
__host__ __device__ constexpr int tall_lds_bytes(int wg_per_cu) { return 160 * 1024 / wg_per_cu - 512; }
constexpr int TALL_LEAN_ARENA = (tall_lds_bytes(TALL_LEAN_WGS) - 1408) & ~15;   // what its workgroup can stage (re_solve_tall.hip asserts it)
// LDS bytes of an entity that stays resident in a tall workgroup: accumulators [nw][d + 1][S + 1], entries, row pointers, labels,
// offsets, weights (the carve-up of re_solve_tall_kernel)
__host__ __device__ inline size_t tall_resident_bytes(int nw, int S, int d, int n, int nnz, bool has_w) {
  return (size_t)nw * (d + 1) * (S + 1) * 8 + (size_t)8 * (nnz + 8) + (size_t)4 * (n + 2) + (size_t)(has_w ? 12 : 8) * n + 16;
}
enum { TALL_VARIANT_LARGE = 0, TALL_VARIANT_SMALL = 1, TALL_VARIANT_LEAN = 2, TALL_VARIANT_TEAM = 3, TALL_VARIANT_MID = 4, TALL_VARIANTS = 5 };
constexpr int BLOCK_NW = 4;   // wavefronts per workgroup of the block kernel
#ifndef GDMIX_TEAM_BLOCK_NW
#define GDMIX_TEAM_BLOCK_NW 8
#endif
constexpr int TEAM_BLOCK_NW = GDMIX_TEAM_BLOCK_NW;   // ... of the compact-form workgroup kernel
constexpr int TEAM_GRID_NW = 8;    // ... of the device-wide kernel (one workgroup per CU)

struct ClassTable {
  int kind[GDMIX_RE_NUM_CLASSES];
  int lds_bytes[GDMIX_RE_NUM_CLASSES];   // LDS bucket of the class; 0 = class disabled (or block class)
  int ncap[GDMIX_RE_NUM_CLASSES];        // quad classes: sample / non-zero capacity of a row's LDS block
  int zcap[GDMIX_RE_NUM_CLASSES];
  int64_t giant_nnz;   // 0 = device-wide kernel off
  int64_t team_nnz;    // 0 = team tiers off
  int tall_min_n;      // entities with p <= TALL_MAX_P and at least this many samples use the tall kernel (0 = never)
  int tall_split_n;    // ... those with at least this many samples one workgroup per CU, the others several
  int tall_adapt_limit;   // > 0: a batch whose eight-wavefront tall class would stay this small with a lower split (2 048, 1 024 or 512 samples)
                          // gets that split (class_base_kernel decides on the device, re_order_kernel moves the entities); 0 = the split is fixed
  int tall_team_n;        // > 0: tall entities of at least this many samples may get a team of workgroups (TALL_T_CLASS); 0 = never
  int tall_team_limit;    // > 0: the class takes the entities above the lowest of tall_team_n x {1, 2, 4} that keeps it within this many
                          // entities (class_base_kernel decides, re_order_kernel moves them); 0 = everything from tall_team_n on
  int tall_mid_n;         // (round 6) > 0: one-wavefront tall entities of at least this many samples go to the mid class whatever the batch
                          // holds (tests); 0: no mid class; < 0: chosen per batch, -tall_mid_n = the class's size limit (one round of its
                          // launch): the lowest of tall_mid_step(k) samples that keeps the class within it, in a small batch only
};
// counts[3 * NUM_CLASSES + k], k = 0..2: one-wavefront tall entities (TALL_S_CLASS) with at least TALL_ADAPT_N[k] samples;
// counts[3 * NUM_CLASSES + TALL_ADAPT_SLOT]: the split class_base_kernel chose (0: none). (The row's team-tier columns hold the tiers' largest entity.)
constexpr int TALL_ADAPT_STEPS = 3;
__host__ __device__ constexpr int tall_adapt_n(int k) { return k == 0 ? 512 : (k == 1 ? 1024 : 2048); }
constexpr int TALL_ADAPT_SLOT = 3;
// counts[3 * NUM_CLASSES + TALL_TEAM_GE + k], k = 0..2: eight-wavefront tall entities with at least tall_team_n << k samples;
// counts[3 * NUM_CLASSES + TALL_TEAM_SLOT]: the threshold class_base_kernel chose (0: no team class in this batch)
constexpr int TALL_TEAM_STEPS = 3;
constexpr int TALL_TEAM_GE = 4;
constexpr int TALL_TEAM_SLOT = 7;
// counts[3 * NUM_CLASSES + TALL_MID_GE + k], k = 0..5: one-wavefront tall entities (TALL_S_CLASS) with at least tall_mid_step(k) samples;
// counts[3 * NUM_CLASSES + TALL_MID_SLOT]: the threshold class_base_kernel chose (0: no mid class in this batch).
// Why a mid class (round 6): a share of a strongly scaled MovieLens job is as long as ONE wavefront needs for the
// largest entity below the per-batch split — a 970-sample user x 49 evaluations, 25 - 40 us each, streamed (it does not fit the eighth
// of a CU's LDS a one-wavefront workgroup has). On four wavefronts with half a CU's LDS the same entity is resident and an
// evaluation is a pass of four samples per lane. Whole populations (thousands of such entities) are bound by throughput, where
// one wavefront per entity is the better use of a CU: the class exists only in a batch small enough for the split to adapt too.
constexpr int TALL_MID_STEPS = 6;
__host__ __device__ constexpr int tall_mid_step(int k) { return k == 0 ? 256 : (k == 1 ? 384 : (k == 2 ? 512 : (k == 3 ? 768 : (k == 4 ? 1024 : 1536)))); }
constexpr int TALL_MID_GE = 8;
constexpr int TALL_MID_SLOT = 14;

// Device pointers of a packed batch, passed by value to kernels.
struct BatchDev {
  const int64_t* ent_row_ptr;
  const int64_t* ent_nnz_ptr;
  const int64_t* ent_feat_ptr;
  const int32_t* row_ptr;
  const int32_t* csr_col;
  const float* csr_val;
  const int32_t* col_ptr;
  const int32_t* csc_row;
  const float* csc_val;
  const float* y;
  const float* offset;
  const float* weight;
  const int32_t* order;
};

struct OutDev {
  double* theta;
  double* theta_thr;
  double* variance;
  double* fval;
  double* gnorm;
  int32_t* nit;
  int32_t* nfev;
  int32_t* status;
};

struct gdmix_ctx_impl {
  int device;
  int num_cus;
  void* scratch;
  size_t scratch_bytes;
  int32_t* host_pinned;   // small pinned buffer for count read-backs (HOST_PINNED_BYTES: the read-back areas, then the mailboxes)
  uint32_t mail_seq[4];   // fetch_small: the last sequence number sent to each mailbox
  int wave_lds_limit;     // entities above this LDS footprint use the block kernel
  int kernel_mask;        // bit1 LDS wave kernel, bit2 group kernels (bit0: the removed register wave kernel, ignored)
  int timing;             // bracket class launches with events
  int64_t giant_nnz;      // entities with >= this many non-zeros use the device-wide kernel (0 = never)
  int64_t team_nnz;       // lowest tier of the team kernel (0 = never)
  int tall_min_n;         // tall kernel for p <= 64 and n >= this (0 = never)
  int tall_split_n;       // tall entities with n >= this: one large workgroup per CU
  int tall_adapt_limit;   // ClassTable::tall_adapt_limit of this device (1.5 x its CUs; GDMIX_RE_TALL_ADAPT overrides, 0 = off)
  int tall_split_set;     // gdmix_re_set_tall_split_n was called: the caller's split is kept, no per-batch adaptation (also when it is the default value)
  int tall_team_n;        // ClassTable::tall_team_n (gdmix_re_set_tall_team_n; GDMIX_RE_TALL_TEAM=0 switches the class off)
  int tall_team_limit;    // ClassTable::tall_team_limit: one round of teams on this device
  int tall_mid_n;         // ClassTable::tall_mid_n (gdmix_re_set_tall_mid_n; GDMIX_RE_TALL_MID=0 switches the class off)
  int spread;             // > 1: large classes are dealt over this many queues (the caller's stream + side streams); 0: one after another
  void* grid_sync;        // device: TeamSync of the team kernels (the first three also: ticket counters of the tall variants), followed
                          // by TALL_TAIL_BYTES for each tall variant (four: the team variant last) and TALL_TEAM_BYTES
  void* big_tmp;          // device: grow-only temporary of the big-entity pack path
  size_t big_tmp_bytes;
  // classes too small to fill the device run on side streams, next to the large ones on the caller's stream and next to each other
  // (round-robin; n_side == 0: off). A share of a strongly scaled job is ALL small classes: on one side stream they ran one after another.
  static constexpr int MAX_SIDE = 4;
  hipStream_t side[MAX_SIDE];
  int n_side;
  hipEvent_t side_fork, side_join[MAX_SIDE];
  hipEvent_t aux_ev[2];   // pack_big_entities: the row table is built on a second stream, next to the column passes
  // gdmix_re_set_defer_unique: the last kernel of a pack (the compaction of the unique feature ids, which no solve kernel reads) runs on
  // the last side stream, next to the solve that follows; `unique_ev` marks its end, `unique_pending` that nobody has waited for it yet
  int defer_unique;
  bool unique_pending;
  hipEvent_t unique_ev;
  hipEvent_t ev0[GDMIX_RE_NUM_CLASSES], ev1[GDMIX_RE_NUM_CLASSES];
  bool ev_used[GDMIX_RE_NUM_CLASSES];
};

// A few bytes from the device to the host WITHOUT a stream synchronise (round 5). The counts a pack or a solve decides its next
// launches on used to come back by hipMemcpyAsync + hipStreamSynchronize: 40 - 140 us each between the producing kernel's end and
// the next launch (the kernel timeline of a MovieLens step, tools/timeline_session.sh: the tiers of a pack started 136 us after the
// 17 us kernel that counts their entities) — a sixth of a strongly scaled share's 1.8 ms step in four such round trips. Now a
// one-wavefront kernel on the same stream copies them into a mailbox in page-locked host memory and sets its flag with a
// system-scope release; the host spins on the flag (FETCH_SPIN_US, then falls back to hipStreamSynchronize: a wait behind 70 ms
// of kernels is not spun through). Stream order makes the flag mean what the synchronise meant for the caller: everything
// queued on `s` before the call is done. GDMIX_RE_MAILBOX=0: the copy + synchronise as before (A/B).
constexpr size_t HOST_PINNED_BYTES = 8192;
constexpr size_t MAILBOX_OFFSET = 4096, MAILBOX_BYTES = 1024, MAILBOX_DATA = 64;   // per box: flag word at 0, data from byte 64
constexpr int FETCH_SPIN_US = 400;
struct SmallFetch { int box; uint32_t seq; size_t bytes; void* host_dst; hipStream_t s; };
hipError_t post_small(gdmix_ctx_impl* ctx, int box, const void* dev_src, size_t bytes, void* host_dst, hipStream_t s, SmallFetch* f);   // queue the hand-over
hipError_t wait_small(gdmix_ctx_impl* ctx, const SmallFetch& f);                                                                    // ... and take it
inline hipError_t fetch_small(gdmix_ctx_impl* ctx, int box, const void* dev_src, size_t bytes, void* host_dst, hipStream_t s) {
  SmallFetch f;
  hipError_t rc = post_small(ctx, box, dev_src, bytes, host_dst, s, &f);
  return rc != hipSuccess ? rc : wait_small(ctx, f);
}

}  // namespace gdmix

// the opaque context of the C ABI
struct gdmix_re_ctx {
  gdmix::gdmix_ctx_impl impl;
};

namespace gdmix {

// LDS bytes the wave kernel needs for an entity of this shape (must match the kernel's carve-up).
__host__ __device__ inline size_t wave_lds_bytes(int p, int n, int nnz, int d, int m, bool has_w) {
  size_t dbl = (size_t)(5 + 2 * m) * p + n + 2 * m;
  size_t w32 = (size_t)4 * nnz + (n + 1) + (d + 1) + (has_w ? 3 : 2) * (size_t)n;
  return ((dbl * 8 + w32 * 4) + 15) & ~(size_t)15;
}

// doubles of global scratch one block-kernel slot needs
inline size_t block_slot_doubles(int64_t max_p, int64_t max_n, int m) {
  // x g d t r | history (tiles of 64 coefficients, re_lbfgs_compact.hpp) | alpha rho | partial sums of split columns | residuals
  return (size_t)5 * max_p + (size_t)2 * m * ((max_p + 63) & ~(int64_t)63) + max_n + 2 * m + 8 + 256 * 64;
}

// Once side streams have forked from the caller's stream, EVERY way out of the function joins them back (an error between
// fork and join used to return with side-stream kernels still running on buffers the caller may then free or reuse).
struct SideJoin {
  gdmix_ctx_impl* ci; hipStream_t main; unsigned used = 0;   // bit k: side stream k has work of this call
  // side stream k joins in: it waits for the fork event (recorded on the caller's stream by the function that owns this object)
  hipError_t use(int k) {
    if (used & (1u << k)) return hipSuccess;
    const hipError_t rc = hipStreamWaitEvent(ci->side[k], ci->side_fork, 0);
    if (rc == hipSuccess) used |= 1u << k;
    return rc;
  }
  void join() {
    for (int k = 0; k < ci->n_side; ++k) {
      if (!(used & (1u << k))) continue;
      if (hipEventRecord(ci->side_join[k], ci->side[k]) != hipSuccess || hipStreamWaitEvent(main, ci->side_join[k], 0) != hipSuccess)
        (void)hipStreamSynchronize(ci->side[k]);   // last resort: the caller's stream must not run ahead of a side stream
    }
    used = 0;
  }
  ~SideJoin() { join(); }
};

hipError_t launch_classify(const gdmix_re_packed* b, int ic, int m, const ClassTable& tab, int32_t* cls_tmp,
                           int32_t* counts_dev, hipStream_t s);
hipError_t launch_order(const gdmix_re_packed* b, int32_t* cls_tmp, const int32_t* class_base_dev,
                        int32_t* cursor_dev, hipStream_t s);
hipError_t launch_solve_quad(int g, int epl, const BatchDev& B, const OutDev& O, const SolveParams& o, const double* theta0,
                             int begin, int count, int ncap, int zcap, hipStream_t s);
hipError_t launch_solve_wave(const BatchDev& B, const OutDev& O, const SolveParams& o, const double* theta0,
                             int begin, int count, int lds_bytes, hipStream_t s);
hipError_t launch_solve_block(const BatchDev& B, const OutDev& O, const SolveParams& o, const double* theta0,
                              int begin, int count, double* scratch, size_t slot_doubles, int slots,
                              int64_t max_p, hipStream_t s);
hipError_t launch_solve_grid(const BatchDev& B, const OutDev& O, const SolveParams& o, const double* theta0,
                             int begin, int count, double* scratch, size_t slot_doubles, int64_t max_p,
                             void* sync_buf, int blocks, int teams, hipStream_t s);
constexpr int TALL_TAIL_BYTES = 256;   // device buffer of the context: padded copy of the end of the batch's row-major arrays
hipError_t launch_solve_tall(int variant, const BatchDev& B, const OutDev& O, const SolveParams& o, const double* theta0, int begin, int count,
                             int num_cus, int64_t Z, void* tail_buf, void* sync_buf, int front_small, hipStream_t s);
hipError_t launch_solve_tall_team(const BatchDev& B, const OutDev& O, const SolveParams& o, const double* theta0, int begin, int count,
                                  int num_cus, int64_t Z, void* tail_buf, void* team_buf, int xcd_fast, hipStream_t s);
void launch_sort_class(int32_t* list, int count, const int64_t* ent_nnz_ptr, hipStream_t s);
hipError_t launch_variance_full(const BatchDev& B, int64_t E, const SolveParams& o, const double* theta, double* variance,
                                double* scratch, size_t slot_doubles, int slots, int64_t max_p, hipStream_t s);
constexpr int64_t VAR_FULL_MAX_P = 2048;   // FULL variance densifies p x p (as the reference does): one wavefront per entity up to here,
constexpr int64_t VAR_FULL_BIG_MAX_P = 16384;   // one entity at a time on the whole device up to here (re_variance_big.hip)
constexpr int VAR_BIG_BUILD_GROUPS = 128;  // workgroups building the Hessian of a large entity, a sample-length vector each
size_t var_full_big_doubles(int64_t max_p, int64_t max_n);
hipError_t launch_variance_full_big(gdmix_ctx_impl* ci, const BatchDev& B, int64_t E, const SolveParams& o, const double* theta,
                                    double* variance, double* scratch, int64_t max_p, int64_t max_n, hipStream_t s);
size_t hessian_dense_scratch_doubles(int64_t n);
hipError_t launch_hessian_dense(gdmix_ctx_impl* ci, const BatchDev& B, int64_t n, int64_t d, int ic, const double* theta, double* H, int64_t ld,
                                double* scratch, hipStream_t s);
hipError_t launch_variance_of_hessian(gdmix_ctx_impl* ci, double* H, double* M, int64_t p, int64_t ld, double l2, int64_t unreg, double* variance,
                                      hipStream_t s);
inline size_t var_full_slot_doubles(int64_t max_p) { return (size_t)2 * max_p * max_p + max_p + 8; }
hipError_t launch_score(const BatchDev& B, int64_t E, int64_t N, int ic, const double* theta, const uint8_t* has_model,
                        float* logit, float* per_coord, hipStream_t s);

// the caller's stream waits for a deferred compaction that is still outstanding (every entry point that reads unique_global, or that
// reuses the workspace it is written from, calls this first; gdmix_re_solve calls it last)
// Every stream that asks waits for the event until the compaction has actually FINISHED (the flag used to be cleared by the
// first caller's wait, so a later reader on another stream was not ordered behind the kernel). A context is single-threaded by
// contract (include/gdmix_re.h: "calls on one context must be serialised by the caller") — a PackedBatch dropped on another thread
// is joined by solver.py under the solver's lock.
inline hipError_t join_unique(gdmix_ctx_impl* ci, hipStream_t s) {
  if (!ci->unique_pending) return hipSuccess;
  const hipError_t q = hipEventQuery(ci->unique_ev);
  if (q == hipSuccess) { ci->unique_pending = false; return hipSuccess; }
  if (q != hipErrorNotReady) return q;
  return hipStreamWaitEvent(s, ci->unique_ev, 0);
}

// pack (re_pack.hip)
size_t pack_workspace_bytes(int64_t E, int64_t N, int64_t Z);
int pack_impl(gdmix_ctx_impl* ctx, const gdmix_re_raw_batch* raw, int has_intercept, void* ws, size_t ws_bytes,
              gdmix_re_packed* out, hipStream_t s);
// pack of the entities too large for the wavefront pack kernels (re_pack_big.hip)
struct BigPackArgs {
  const int64_t* ent_row_ptr;
  const int64_t* ent_nnz_ptr;
  const int64_t* row_nnz_ptr;
  const int64_t* col_global;
  const float* val;
  int ic;
  int32_t* row_ptr;
  int32_t* csr_col;
  int32_t* col_ptr;
  int32_t* csc_row;
  float* csc_val;
  int32_t* uniq_sparse;
  int32_t* d_cnt;
  const int32_t* big_list;
  int n_big;
  int64_t big_nnz;
  int* max_p;   // device
  int* err;     // device
};
int pack_big_entities(gdmix_ctx_impl* ctx, const BigPackArgs& a, hipStream_t s, hipStream_t aux);

hipError_t launch_partition_ids(const int64_t* ids, int64_t count, int32_t num_partitions, int32_t* out,
                                hipStream_t s);

void set_error(const char* fmt, ...);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: one flag per call site would leave the
// second device of a process at the 64 KB default. One of these per call site (static), a bit per device.
// Persistent grids (the multi-workgroup team tiers, the device-wide kernel, the tall teams) need ALL their workgroups resident at
// once — they meet at barriers — and one such grid takes a whole CU per workgroup (512 threads at two wavefronts per SIMD, or a
// whole CU's LDS). Two of them started at the same time from two contexts of a process (a host pipeline over several contexts:
// model.py, bench.py's hand-over and full-share legs) can each get half the device and wait for the other half until the
// barrier's watchdog gives up (round 5: 2 of 12.5 M entities ABORTED, 5 s lost, with three contexts on Zipf-sized partitions).
// The gate chains them per device: a grid's launch waits (on the device, stream-ordered: the host does not block) for the
// previous grid of ANY context of this process to finish. ScopedGridGate: construct before the launch, destroy after it.
// Between PROCESSES on one device (round 6) the same chain is a file lock per device, held while a process has such a grid in
// flight (GridLock, re_api.hip). Gated: the team tiers and the device-wide kernel (re_solve.hip), the tall teams (re_solve_tall.hip).
// NOT gated: fe_tail_kernel (fe_solve.hip) — its launches are queued ahead of collectives that wait for the other workers, so a lock
// held for queued work could deadlock two fixed-effect workers on one device; gdmix_fe_create falls back to the three-launch step
// (no workgroup waits for another) when another process is present on the device — and the pack tiers, which hold no workgroup
// waiting for another.
// This process is (now) present on the device / another process is: record locks on a per-device file (re_api.hip).
void device_register_process(int device);
bool device_has_another_process(int device);

class ScopedGridGate {
 public:
  ScopedGridGate(int device, hipStream_t s);
  ~ScopedGridGate();
  hipError_t status() const { return err_; }
 private:
  int device_;
  hipStream_t s_;
  hipError_t err_;
  void* held_;      // the inter-process lock of the device (GridLock, re_api.hip) when this launch holds a count of it
};

struct DynLdsOnce {
  unsigned long long done = 0;   // (set twice by racing threads at worst: the call is idempotent)
  hipError_t set(const void* fn, int bytes = 160 * 1024) {
    int dev = 0;
    hipError_t rc = hipGetDevice(&dev);
    if (rc != hipSuccess) return rc;
    if (dev >= 0 && dev < 64 && ((done >> dev) & 1ull)) return hipSuccess;
    rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (rc == hipSuccess && dev >= 0 && dev < 64) done |= 1ull << dev;
    return rc;
  }
};

}  // namespace gdmix
