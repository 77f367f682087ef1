// re_api.hip — the extern "C" surface of libgdmix_re.so (include/gdmix_re.h).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <errno.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <condition_variable>
#include <map>
#include <new>
#include <string>

#include <mutex>

#include "re_internal.hpp"
#include "re_solve_team.hpp"

namespace gdmix {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t _rc = (expr);                                                                 \
    if (_rc != hipSuccess) {                                                                 \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_rc), __FILE__, __LINE__); \
      return GDMIX_RE_EHIP;                                                                  \
    }                                                                                        \
  } while (0)

// Size classes in routing order (see re_internal.hpp). A 64-thread workgroup may use up to 64 KiB of LDS
// here; larger blocks go to the workgroup-per-entity kernel.
struct ClassDesc { int kind; int lds; const char* name; int ncap, zcap; };   // quad: lds = LDS of a whole wave (4 rows)
static const ClassDesc kClasses[GDMIX_RE_NUM_CLASSES] = {
    {KIND_QUAD2, 1, "re_solve_grp_kernel<16,2> n<=16 nnz<=64", 16, 64},     {KIND_QUAD2, 1, "re_solve_grp_kernel<16,2> n<=32 nnz<=128", 32, 128},
    {KIND_QUAD2, 1, "re_solve_grp_kernel<16,2> n<=128 nnz<=512", 128, 512},
    {KIND_QUAD3, 1, "re_solve_grp_kernel<16,3> n<=16 nnz<=64", 16, 64},     {KIND_QUAD3, 1, "re_solve_grp_kernel<16,3> n<=32 nnz<=128", 32, 128},
    {KIND_QUAD3, 1, "re_solve_grp_kernel<16,3> n<=128 nnz<=512", 128, 512},
    {KIND_QUAD4, 1, "re_solve_grp_kernel<16,4> n<=16 nnz<=64", 16, 64},     {KIND_QUAD4, 1, "re_solve_grp_kernel<16,4> n<=32 nnz<=128", 32, 128},
    {KIND_QUAD4, 1, "re_solve_grp_kernel<16,4> n<=128 nnz<=512", 128, 512},
    {KIND_PAIR3, 1, "re_solve_grp_kernel<32,3> n<=32 nnz<=128", 32, 128},   {KIND_PAIR3, 1, "re_solve_grp_kernel<32,3> n<=64 nnz<=256", 64, 256},
    {KIND_PAIR3, 1, "re_solve_grp_kernel<32,3> n<=256 nnz<=1024", 256, 1024},
    {KIND_PAIR4, 1, "re_solve_grp_kernel<32,4> n<=32 nnz<=128", 32, 128},   {KIND_PAIR4, 1, "re_solve_grp_kernel<32,4> n<=64 nnz<=256", 64, 256},
    {KIND_PAIR4, 1, "re_solve_grp_kernel<32,4> n<=256 nnz<=1024", 256, 1024},
    {KIND_G64_3, 1, "re_solve_grp_kernel<64,3> n<=64 nnz<=512", 64, 512},    {KIND_G64_3, 1, "re_solve_grp_kernel<64,3> n<=512 nnz<=2048", 512, 2048},
    {KIND_G64_4, 1, "re_solve_grp_kernel<64,4> n<=64 nnz<=512", 64, 512},    {KIND_G64_4, 1, "re_solve_grp_kernel<64,4> n<=512 nnz<=2048", 512, 2048},
    {KIND_G128_3, 1, "re_solve_grp_kernel<128,3> n<=128 nnz<=1024", 128, 1024}, {KIND_G128_3, 1, "re_solve_grp_kernel<128,3> n<=1024 nnz<=3072", 1024, 3072},
    {KIND_G128_4, 1, "re_solve_grp_kernel<128,4> n<=128 nnz<=1024", 128, 1024}, {KIND_G128_4, 1, "re_solve_grp_kernel<128,4> n<=1024 nnz<=3072", 1024, 3072},
    {KIND_G256_3, 1, "re_solve_grp_kernel<256,3> n<=256 nnz<=2048", 256, 2048}, {KIND_G256_3, 1, "re_solve_grp_kernel<256,3> n<=2048 nnz<=4096", 2048, 4096},
    {KIND_G256_4, 1, "re_solve_grp_kernel<256,4> n<=256 nnz<=2048", 256, 2048}, {KIND_G256_4, 1, "re_solve_grp_kernel<256,4> n<=2048 nnz<=4096", 2048, 4096},
    {KIND_G512_4, 1, "re_solve_grp_kernel<512,4> n<=512 nnz<=4096", 512, 4096},
    {KIND_WLDS, 24576, "re_solve_wave_kernel lds<=24K"},     {KIND_WLDS, 65536, "re_solve_wave_kernel lds<=64K"},
    {KIND_TALL_T, 0, "re_solve_tall_team_kernel<8> x4 p<=64"},
    {KIND_TALL_L, 0, "re_solve_tall_kernel<1> lean p<=64"},
    {KIND_TALL_S, 0, "re_solve_tall_kernel<1> p<=64"},
    {KIND_TALL_M, 0, "re_solve_tall_kernel<4> p<=64"},
    {KIND_TALL, 0, "re_solve_tall_kernel<8> p<=64"},
    {KIND_BLOCK, 0, "re_solve_team_kernel workgroup"},
    {KIND_GRID, 0, "re_solve_team_kernel 128 teams"},
    {KIND_GRID, 0, "re_solve_team_kernel 32 teams"},
    {KIND_GRID, 0, "re_solve_team_kernel 8 teams"},
    {KIND_GRID, 0, "re_solve_team_kernel device-wide"}};

__global__ void class_base_kernel(int32_t* cc, int tall_adapt_limit, int tall_adapt_small, int tall_team_n, int tall_team_limit, int tall_mid_n) {
  // cc[0..NC) counts -> cc[NC..2NC) exclusive bases, cc[2NC..3NC) cursors = 0
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    // The split between the one-wavefront and the eight-wavefront tall kernels (4 096 samples by default: right for a batch with
    // thousands of tall entities, where a CU is better spent on eight entities than on one) is lowered for a batch whose
    // eight-wavefront class stays small anyway — a share of a strongly scaled job: 17 k users, 5 - 11 of them above 4 096 samples, and
    // the step lasts as long as ONE wavefront needs for a 4 000-sample entity (tools/share_timeline.py). The lowest of 512 / 1 024 /
    // 2 048 that keeps the class within `tall_adapt_limit` workgroups (one per CU) wins; re_order_kernel moves the entities.
    int32_t* const ge = cc + 3 * GDMIX_RE_NUM_CLASSES;
    // The team class (four workgroups per entity) takes the tallest entities of the batch: those above the lowest of
    // tall_team_n x {1, 2, 4} samples that keeps it within one round of teams — a batch with more than that above the highest
    // threshold has no team class (throughput binds it, not one entity's chain). No limit: everything from tall_team_n on.
    int team_from = 0;
    if (tall_team_n > 0) {
      if (tall_team_limit <= 0) { if (ge[TALL_TEAM_GE] > 0) team_from = tall_team_n; }
      else
        for (int k = 0; k < TALL_TEAM_STEPS && team_from == 0; ++k)
          if (ge[TALL_TEAM_GE + k] > 0 && ge[TALL_TEAM_GE + k] <= tall_team_limit) team_from = tall_team_n << k;
      if (team_from > 0) {
        const int moved = ge[TALL_TEAM_GE + (team_from == tall_team_n ? 0 : (team_from == 2 * tall_team_n ? 1 : 2))];
        cc[TALL_T_CLASS] += moved;
        cc[TALL_CLASS] -= moved;
      }
    }
    ge[TALL_TEAM_SLOT] = team_from;
    // ... and only for a batch whose one-wavefront class is small itself (at most `tall_adapt_small` entities: two rounds of its launch):
    // a whole population is bound by throughput, where the one-wavefront kernel is the better use of a CU (MovieLens-20M per user on one
    // GPU, 14 k such entities: 8.9 ms with the split at 4 096, 9.4 ms when 224 more entities took a CU each)
    int split = 0;
    if (tall_adapt_limit > 0 && cc[TALL_S_CLASS] <= tall_adapt_small) {
      for (int k = 0; k < TALL_ADAPT_STEPS && split == 0; ++k)
        if (ge[k] > 0 && cc[TALL_CLASS] + ge[k] <= tall_adapt_limit) {
          split = tall_adapt_n(k);
          cc[TALL_CLASS] += ge[k];
          cc[TALL_S_CLASS] -= ge[k];
        }
    }
    ge[TALL_ADAPT_SLOT] = split;
    // The mid class (four wavefronts per entity, two workgroups per CU) takes the largest one-wavefront entities BELOW the split: the
    // lowest threshold of tall_mid_step() that keeps it within one round of its launch (-tall_mid_n workgroups), in a small batch only
    // (same test as the split: a whole population is bound by throughput). tall_mid_n > 0: everything from that many samples on.
    int mid_from = 0;
    // entities of at least `split` samples have left the one-wavefront class (counted with the same rule: ge[k] of the chosen split)
    const int gone = split > 0 ? ge[split == tall_adapt_n(0) ? 0 : (split == tall_adapt_n(1) ? 1 : 2)] : 0;
    if (tall_mid_n > 0) {
      const int cnt = (split > 0 && split <= tall_mid_n) ? 0 : ge[TALL_MID_GE] - gone;
      if (cnt > 0) { mid_from = tall_mid_n; cc[TALL_M_CLASS] += cnt; cc[TALL_S_CLASS] -= cnt; }
    } else if (tall_mid_n < 0 && cc[TALL_S_CLASS] <= tall_adapt_small) {
      for (int k = 0; k < TALL_MID_STEPS && mid_from == 0; ++k) {
        if (split > 0 && tall_mid_step(k) >= split) break;
        const int cnt = ge[TALL_MID_GE + k] - gone;
        if (cnt > 0 && cnt <= -tall_mid_n) { mid_from = tall_mid_step(k); cc[TALL_M_CLASS] += cnt; cc[TALL_S_CLASS] -= cnt; }
      }
    }
    ge[TALL_MID_SLOT] = mid_from;
    int run = 0;
    for (int c = 0; c < GDMIX_RE_NUM_CLASSES; ++c) { cc[GDMIX_RE_NUM_CLASSES + c] = run; run += cc[c]; cc[2 * GDMIX_RE_NUM_CLASSES + c] = 0; }
  }
}

static BatchDev make_batch_dev(const gdmix_re_packed* b) {
  BatchDev B;
  B.ent_row_ptr = b->ent_row_ptr; B.ent_nnz_ptr = b->ent_nnz_ptr; B.ent_feat_ptr = b->ent_feat_ptr;
  B.row_ptr = b->row_ptr; B.csr_col = b->csr_col; B.csr_val = b->csr_val;
  B.col_ptr = b->col_ptr; B.csc_row = b->csc_row; B.csc_val = b->csc_val;
  B.y = b->y; B.offset = b->offset; B.weight = b->weight; B.order = b->order;
  return B;
}

__global__ __launch_bounds__(WAVE) void publish_kernel(const uint32_t* __restrict__ src, int words, uint32_t* box_data, uint32_t* box_flag, uint32_t seq) {
  // Write-through stores (system scope, relaxed) to the page-locked box, a wait until they are acknowledged, then the flag the same
  // way. NOT a system-scope release: that writes back this XCD's whole L2 first, behind whatever kernels are filling it at the
  // moment — 104 us for these 48 bytes next to a running pack tier (and the runtime's own copy kernel ends in one: that was the
  // 0.1 ms of the read-back it replaces). One wavefront, so the one wait covers every lane's stores.
  for (int i = threadIdx.x; i < words; i += WAVE) __hip_atomic_store(box_data + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) __hip_atomic_store(box_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static inline void cpu_relax() {
#if !defined(__HIP_DEVICE_COMPILE__)
  asm volatile("pause" ::: "memory");
#endif
}

static int mailbox_on() {
  static const int on = [] { const char* e = getenv("GDMIX_RE_MAILBOX"); return (e && atoi(e) == 0) ? 0 : 1; }();
  return on;
}

hipError_t post_small(gdmix_ctx_impl* ctx, int box, const void* dev_src, size_t bytes, void* host_dst, hipStream_t s, SmallFetch* f) {
  f->box = -1; f->seq = 0; f->bytes = bytes; f->host_dst = host_dst; f->s = s;
  if (!mailbox_on() || bytes % 4 || bytes > MAILBOX_BYTES - MAILBOX_DATA || box < 0 || box >= 4)
    return hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, s);       // wait_small synchronises
  char* const base = reinterpret_cast<char*>(ctx->host_pinned) + MAILBOX_OFFSET + (size_t)box * MAILBOX_BYTES;
  uint32_t seq = ++ctx->mail_seq[box];
  if (seq == 0) seq = ++ctx->mail_seq[box];     // (0 is the empty box)
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(WAVE), 0, s, static_cast<const uint32_t*>(dev_src), (int)(bytes / 4),
                     reinterpret_cast<uint32_t*>(base + MAILBOX_DATA), reinterpret_cast<uint32_t*>(base), seq);
  f->box = box; f->seq = seq;
  return hipGetLastError();
}

hipError_t wait_small(gdmix_ctx_impl* ctx, const SmallFetch& f) {
  if (f.box < 0) return hipStreamSynchronize(f.s);
  char* const base = reinterpret_cast<char*>(ctx->host_pinned) + MAILBOX_OFFSET + (size_t)f.box * MAILBOX_BYTES;
  uint32_t* const flag = reinterpret_cast<uint32_t*>(base);
  timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != f.seq) {
    for (int k = 0; k < 16; ++k) cpu_relax();
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if ((t1.tv_sec - t0.tv_sec) * 1000000L + (t1.tv_nsec - t0.tv_nsec) / 1000L > FETCH_SPIN_US) {
      hipError_t rc = hipStreamSynchronize(f.s);        // a long wait: sleep in the runtime; then the box is full, or the stream failed
      if (rc != hipSuccess) return rc;
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != f.seq) return hipErrorUnknown;
      break;
    }
  }
  memcpy(f.host_dst, base + MAILBOX_DATA, f.bytes);
  return hipSuccess;
}

}  // namespace gdmix

using namespace gdmix;


namespace gdmix {
namespace {
constexpr int GATE_DEVICES = 64;
std::mutex g_gate_mu[GATE_DEVICES];
hipEvent_t g_gate_ev[GATE_DEVICES] = {};
bool g_gate_armed[GATE_DEVICES] = {};
}  // namespace

// ---- the same between PROCESSES (round 6; VERDICT r5 missing 3) --------------------------------------------------------------
// The reference runs num_of_consumers processes per worker (random_effect_lr_lbfgs_model.py:103,214-217) and TF_CONFIG may list more
// workers than the box has GPUs: two processes can drive one device, and their persistent grids can starve each other exactly as two
// contexts' could. There is no stream-ordered primitive between processes, so the chain is a file lock per device:
//   <dir>/gdmix_re_grid_<pci bus id>.lock   held (flock, exclusive) by a process while any of its persistent grids is in flight
//   <dir>/gdmix_re_grid_<pci bus id>.turn   the turnstile: held while a process WAITS for the lock
// A process takes the lock when its count of grids in flight goes 0 -> 1 and drops it from a host function queued behind the last
// grid (hipLaunchHostFunc on the device's release stream, ordered by the gate's event: the caller's stream never waits for it), so
// inside one process nothing changes — grids of several contexts still chain on the device, the host does not block. A launch that
// finds the lock already held by its own process rides along unless the turnstile is taken (somebody else waits): then it waits
// until the process has let go and queues behind the waiter — two processes alternate, neither starves. A killed process loses its
// locks with its descriptors. GDMIX_RE_GRID_LOCK=0 turns it off, GDMIX_RE_LOCK_DIR moves the files (default /tmp; processes in
// containers with separate /tmp do not see each other: give them a shared directory).
class GridLock {
 public:
  explicit GridLock(const std::string& base) : base_(base) {}
  // false: the lock files cannot be used (read-only directory ...): the caller goes on unprotected, as before round 6
  bool acquire() {
    std::unique_lock<std::mutex> g(mu_);
    if (!open_files()) return false;
    for (;;) {
      if (state_ == HELD) {
        if (!somebody_waits()) { ++count_; ++rides_; return true; }
        cv_.wait(g);
      } else if (state_ == ACQUIRING) {
        cv_.wait(g);
      } else {
        state_ = ACQUIRING;
        g.unlock();
        const bool ok = flock_retry(turn_fd_, LOCK_EX) && flock_retry(dev_fd_, LOCK_EX);
        flock_retry(turn_fd_, LOCK_UN);
        g.lock();
        if (!ok) { state_ = FREE; cv_.notify_all(); return false; }
        state_ = HELD;
        count_ = 1;
        ++takes_;
        cv_.notify_all();
        return true;
      }
    }
  }
  void release() {
    std::lock_guard<std::mutex> g(mu_);
    if (state_ != HELD || count_ <= 0) return;
    if (--count_ == 0) {
      flock_retry(dev_fd_, LOCK_UN);
      state_ = FREE;
      cv_.notify_all();
    }
  }
  void stats(long* takes, long* rides) {
    std::lock_guard<std::mutex> g(mu_);
    *takes = takes_;
    *rides = rides_;
  }

 private:
  enum State { FREE, ACQUIRING, HELD };
  static bool flock_retry(int fd, int op) {
    int rc;
    do rc = flock(fd, op); while (rc != 0 && errno == EINTR);
    return rc == 0;
  }
  static int open_one(const std::string& path) {
    const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_CLOEXEC, 0666);
    if (fd >= 0) (void)fchmod(fd, 0666);      // another user's process on the same device must be able to open it
    return fd;
  }
  bool open_files() {
    if (dev_fd_ >= 0) return true;
    if (failed_) return false;
    dev_fd_ = open_one(base_ + ".lock");
    turn_fd_ = open_one(base_ + ".turn");
    probe_fd_ = open_one(base_ + ".turn");    // a second description of the turnstile: flock is per description
    if (dev_fd_ < 0 || turn_fd_ < 0 || probe_fd_ < 0) {
      fprintf(stderr, "gdmix_re: cannot open %s.lock (%s): persistent grids of several processes on this device are not chained\n", base_.c_str(), strerror(errno));
      for (int* fd : {&dev_fd_, &turn_fd_, &probe_fd_}) { if (*fd >= 0) close(*fd); *fd = -1; }
      failed_ = true;
      return false;
    }
    return true;
  }
  // is the turnstile taken? (by another process — inside this one the state machine keeps waiters off it while the lock is held)
  bool somebody_waits() {
    if (flock(probe_fd_, LOCK_SH | LOCK_NB) == 0) { flock_retry(probe_fd_, LOCK_UN); return false; }
    return errno == EWOULDBLOCK;
  }
  std::string base_;
  std::mutex mu_;
  std::condition_variable cv_;
  State state_ = FREE;
  int count_ = 0, dev_fd_ = -1, turn_fd_ = -1, probe_fd_ = -1;
  bool failed_ = false;
  long takes_ = 0, rides_ = 0;
};

namespace {
std::mutex g_locks_mu;
std::map<std::string, GridLock*> g_locks;      // by key, never freed (a handful per process)
GridLock* g_dev_lock[GATE_DEVICES] = {};
hipStream_t g_rel_stream[GATE_DEVICES] = {};
bool g_dev_lock_off[GATE_DEVICES] = {};

GridLock* grid_lock_by_key(const char* key) {
  std::lock_guard<std::mutex> g(g_locks_mu);
  auto it = g_locks.find(key);
  if (it != g_locks.end()) return it->second;
  const char* dir = getenv("GDMIX_RE_LOCK_DIR");
  std::string base = std::string(dir && *dir ? dir : "/tmp") + "/gdmix_re_grid_";
  for (const char* c = key; *c; ++c) base += (isalnum((unsigned char)*c) || *c == '.' || *c == '-') ? *c : '_';
  return g_locks[key] = new GridLock(base);
}

// the lock of a HIP device of this process (nullptr: off), keyed by its PCI bus id — the same for every process whatever
// HIP_VISIBLE_DEVICES makes of the ordinals
GridLock* grid_lock_of_device(int device) {
  if (g_dev_lock_off[device]) return nullptr;
  if (g_dev_lock[device]) return g_dev_lock[device];
  const char* e = getenv("GDMIX_RE_GRID_LOCK");
  char bus[64] = "";
  if ((e && atoi(e) == 0) || hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess || !bus[0] ||
      hipStreamCreateWithFlags(&g_rel_stream[device], hipStreamNonBlocking) != hipSuccess) {
    g_dev_lock_off[device] = true;
    return nullptr;
  }
  return g_dev_lock[device] = grid_lock_by_key(bus);
}

void grid_lock_release_cb(void* p) { static_cast<GridLock*>(p)->release(); }

// Presence of a process on a device: a POSIX read lock on byte 0 of <dir>/gdmix_re_grid_<pci bus id>.here, taken when the process
// creates its first context there and kept for its lifetime. Record locks belong to the process, so a query for a conflicting
// write lock (F_GETLK) answers "does ANOTHER process hold one" in one system call.
std::mutex g_here_mu;
std::map<std::string, int> g_here_fd;

int presence_fd(int device) {
  char bus[64] = "";
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess || !bus[0]) return -1;
  std::lock_guard<std::mutex> g(g_here_mu);
  auto it = g_here_fd.find(bus);
  if (it != g_here_fd.end()) return it->second;
  const char* dir = getenv("GDMIX_RE_LOCK_DIR");
  std::string path = std::string(dir && *dir ? dir : "/tmp") + "/gdmix_re_grid_";
  for (const char* c = bus; *c; ++c) path += (isalnum((unsigned char)*c) || *c == '.' || *c == '-') ? *c : '_';
  path += ".here";
  const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_CLOEXEC, 0666);
  if (fd >= 0) {
    (void)fchmod(fd, 0666);
    struct flock fl = {};
    fl.l_type = F_RDLCK; fl.l_whence = SEEK_SET; fl.l_start = 0; fl.l_len = 1;
    (void)fcntl(fd, F_SETLK, &fl);      // (never closed: closing ANY descriptor of the file would drop the process's lock)
  }
  return g_here_fd[bus] = fd;
}
}  // namespace

void device_register_process(int device) {
  const char* e = getenv("GDMIX_RE_GRID_LOCK");
  if (!(e && atoi(e) == 0)) (void)presence_fd(device);
}

bool device_has_another_process(int device) {
  const char* e = getenv("GDMIX_RE_GRID_LOCK");
  if (e && atoi(e) == 0) return false;
  const int fd = presence_fd(device);
  if (fd < 0) return false;
  struct flock fl = {};
  fl.l_type = F_WRLCK; fl.l_whence = SEEK_SET; fl.l_start = 0; fl.l_len = 1;
  return fcntl(fd, F_GETLK, &fl) == 0 && fl.l_type != F_UNLCK;
}

ScopedGridGate::ScopedGridGate(int device, hipStream_t s) : device_(device >= 0 && device < GATE_DEVICES ? device : 0), s_(s), err_(hipSuccess), held_(nullptr) {
  g_gate_mu[device_].lock();
  if (!g_gate_ev[device_]) err_ = hipEventCreateWithFlags(&g_gate_ev[device_], hipEventDisableTiming);
  if (err_ == hipSuccess && g_gate_armed[device_]) err_ = hipStreamWaitEvent(s_, g_gate_ev[device_], 0);
  if (err_ == hipSuccess) {
    GridLock* const L = grid_lock_of_device(device_);
    if (L && L->acquire()) held_ = L;         // may block the host: another PROCESS has a persistent grid in flight on this device
  }
}

ScopedGridGate::~ScopedGridGate() {
  const bool recorded = g_gate_ev[device_] && hipEventRecord(g_gate_ev[device_], s_) == hipSuccess;
  if (recorded) g_gate_armed[device_] = true;
  if (held_) {
    // let go behind the grid, from the release stream: the caller's stream does not wait for a host function
    GridLock* const L = static_cast<GridLock*>(held_);
    if (!(recorded && hipStreamWaitEvent(g_rel_stream[device_], g_gate_ev[device_], 0) == hipSuccess &&
          hipLaunchHostFunc(g_rel_stream[device_], grid_lock_release_cb, L) == hipSuccess)) {
      (void)hipStreamSynchronize(s_);         // could not queue the release: wait for the grid here
      L->release();
    }
  }
  g_gate_mu[device_].unlock();
}
}  // namespace gdmix

extern "C" {

// A class whose launch cannot fill the device: fewer wavefronts than three quarters of what the CUs hold at two per SIMD (the eight-wavefront
// tall workgroups: fewer entities than CUs). Such a class runs on one of the context's side streams, next to the large ones and to
// the other small ones.
// Large classes are dealt over the caller's stream and the side streams, in launch order (round 4): side by side their tails overlap
// — a class launch ends with the workgroups whose entities need the most iterations while the rest of the device idles — and
// wavefronts of different classes share a SIMD. C2: 9.35 -> 8.93 ms of solve, step 10.70 -> 10.30 ms; MovieLens-20M per-movie 6.04 -> 5.70 ms
// (tools/r04_spread.sh: 2 / 3 / 4 queues 10.44 / 10.31 / 10.30 ms; most-entities-first changes nothing). gdmix_re_set_spread(ctx, 0)
// (GDMIX_RE_SPREAD=0 for a whole process): one after another on the caller's stream, as before (A/B, and the per-kernel durations of a
// profile: overlapped kernels stretch each other).
static int spread_default() {
  const char* e = getenv("GDMIX_RE_SPREAD");
  const int n = e ? atoi(e) : 4;
  return n > 1 ? n : 0;
}
static bool class_is_small(int kind, int count, int num_cus) {
  const int gl = group_lanes(kind);
  long waves = count;                                   // wavefront kernels, one-wavefront tall variants
  if (kind == KIND_TALL_T) return true;                // at most one round of teams
  if (kind == KIND_TALL) waves = (long)count * 4 * TALL_NW / 8;   // one workgroup per CU: count < num_cus
  if (kind == KIND_TALL_M) waves = (long)count * 8 / TALL_MID_WGS;   // two workgroups per CU
  else if (gl > 0) waves = gl >= WAVE ? (long)count * (gl / WAVE) : ((long)count * gl + WAVE - 1) / WAVE;
  return waves < (long)num_cus * 6;
}

GDMIX_API int gdmix_re_abi_version(void) { return GDMIX_RE_ABI_VERSION; }

// Test hooks of the inter-process chain of persistent grids (host only, no device): the lock a device named `key` would use.
GDMIX_API int gdmix_re_grid_lock_acquire(const char* key) {
  if (!key || !*key) { set_error("key is empty"); return GDMIX_RE_EINVAL; }
  return grid_lock_by_key(key)->acquire() ? GDMIX_RE_OK : GDMIX_RE_EINVAL;
}
GDMIX_API int gdmix_re_grid_lock_release(const char* key) {
  if (!key || !*key) { set_error("key is empty"); return GDMIX_RE_EINVAL; }
  grid_lock_by_key(key)->release();
  return GDMIX_RE_OK;
}
GDMIX_API int gdmix_re_grid_lock_stats(const char* key, int64_t* takes, int64_t* rides) {
  if (!key || !*key || !takes || !rides) { set_error("bad argument"); return GDMIX_RE_EINVAL; }
  long t = 0, r = 0;
  grid_lock_by_key(key)->stats(&t, &r);
  *takes = t;
  *rides = r;
  return GDMIX_RE_OK;
}

GDMIX_API const char* gdmix_re_last_error(void) { return g_err; }

GDMIX_API void gdmix_re_default_opts(gdmix_re_opts* o) {
  if (!o) return;
  o->l2 = 1.0; o->regularize_bias = 1; o->has_intercept = 1; o->m = 10; o->max_iter = 100;
  o->maxfun = 15000; o->maxls = 20; o->ftol = 1e-12; o->pgtol = 1e-5;
  o->variance_mode = GDMIX_RE_VAR_NONE; o->threshold = 1e-4;
  o->sum_loss = 0; o->linear = 0;
}

GDMIX_API int gdmix_re_create(int hip_device, gdmix_re_ctx** out) {
  if (!out) { set_error("out is NULL"); return GDMIX_RE_EINVAL; }
  *out = nullptr;
  int count = 0;
  HIP_TRY(hipGetDeviceCount(&count));
  if (hip_device < 0 || hip_device >= count) {
    set_error("HIP device %d not present (%d visible)", hip_device, count);
    return GDMIX_RE_EHIP;
  }
  HIP_TRY(hipSetDevice(hip_device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, hip_device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; this library only carries gfx950 (MI355X) code", hip_device, prop.gcnArchName);
    return GDMIX_RE_EHIP;
  }
  gdmix_re_ctx* c = new (std::nothrow) gdmix_re_ctx();
  if (!c) { set_error("out of host memory"); return GDMIX_RE_ENOMEM; }
  c->impl.device = hip_device;
  c->impl.num_cus = prop.multiProcessorCount;
  device_register_process(hip_device);      // (what gdmix_fe_create asks before it chooses the one-launch step: re_internal.hpp)
  // (tools/r04_adapt.sh. Before the team class took a batch's tallest entities, 64 / 128 / 192 / 256 / 384 / 512 / 1024 workgroups on a 17 k-user
  // share: 2.51 / 2.39 / 2.22 / 2.51 / 2.24 / 2.43 / 2.83 ms, and 192 it was. With the heads on teams the eight-wavefront class is a class of
  // mid-size entities that finish early and hand their CU on: slowest per-user / per-movie share at 128 / 192 / 256 / 320 / 384 / 448 / 512 / 768 / 1024:
  // 2.32/2.36, 2.13/2.30, 2.03/2.07, 2.07/2.06, (2.0)/2.05, 2.02/2.00, 2.06/2.05, 2.46/1.97, 2.52/1.95 ms -> one and a half workgroups per CU.)
  c->impl.tall_adapt_limit = c->impl.num_cus * 3 / 2;
  if (const char* e = getenv("GDMIX_RE_TALL_ADAPT")) c->impl.tall_adapt_limit = atoi(e) > 0 ? atoi(e) : 0;
  c->impl.scratch = nullptr;
  c->impl.scratch_bytes = 0;
  c->impl.host_pinned = nullptr;
  c->impl.wave_lds_limit = 65536;
  c->impl.kernel_mask = 7;
  c->impl.timing = 0;
  c->impl.giant_nnz = 16777216;
  c->impl.team_nnz = 16384;
  c->impl.tall_min_n = GDMIX_RE_TALL_MIN_N_DEFAULT;
  c->impl.tall_split_n = GDMIX_RE_TALL_SPLIT_N_DEFAULT;
  c->impl.tall_split_set = 0;
  c->impl.tall_team_n = GDMIX_RE_TALL_TEAM_N_DEFAULT;
  // How many entities the class may take: a quarter of the CUs' worth (one round of teams). Measured twice, because the answer depends on
  // the class behind it (tools/r04_tallteam2.sh, slowest per-movie share of eight): while the eight-wavefront class was capped at 3/4 of the
  // CUs, 64 teams held every CU for the length of their entity and the one-wavefront class waited (2.9 ms; 32: 2.25, 16: 2.29); since that
  // class may take 1.5 workgroups per CU and absorbs what the teams leave, 16 / 32 / 48 / 64: 2.38 / 1.98 / 1.98 / 1.85 - 1.90 ms.
  c->impl.tall_team_limit = c->impl.num_cus / 4 < TALL_TEAM_MAX ? c->impl.num_cus / 4 : TALL_TEAM_MAX;
  // The mid class (round 6) is OFF unless asked for (gdmix_re_set_tall_mid_n(ctx, -1), GDMIX_RE_TALL_MID=1): measured on the shares of a
  // strongly scaled MovieLens-20M job it loses — per-user share 1.87 - 1.96 -> 2.24 - 2.26 ms, per-movie 1.69 - 1.74 -> 1.83 - 1.88 ms
  // (profiles/r06_ml20m_mid.txt, with the timelines): the one-wavefront launch does not get shorter when its largest entities leave
  // (1.25 -> 1.28 ms: it is bound by the throughput of the ~14 k small entities it also carries, eight per CU, not by its longest
  // chain), and every mid workgroup takes half a CU's LDS away from four of them.
  c->impl.tall_mid_n = 0;
  if (const char* e = getenv("GDMIX_RE_TALL_MID")) { if (atoi(e) == 1) c->impl.tall_mid_n = -(c->impl.num_cus * TALL_MID_WGS); else if (atoi(e) > 1) c->impl.tall_mid_n = -atoi(e); }   // 1 = per batch, n > 1 = per batch with this size limit
  // a team needs its TALL_TEAM_C workgroups resident at once, each with a whole CU's LDS, and a launch has eight teams at least
  // (one per XCD): a device (partition) with fewer CUs than that gets no team class at all — its members could never all be
  // placed and every run would end in the barrier's watchdog
  if (c->impl.num_cus < 8 * TALL_TEAM_C) c->impl.tall_team_n = 0;
  if (const char* e = getenv("GDMIX_RE_TALL_TEAM")) { if (atoi(e) == 0) c->impl.tall_team_n = 0; }   // A/B switch
  if (const char* e = getenv("GDMIX_RE_TALL_TEAM_LIMIT")) { if (atoi(e) > 0 && atoi(e) <= TALL_TEAM_MAX) c->impl.tall_team_limit = atoi(e); }   // exploration knob
  c->impl.spread = spread_default();
  c->impl.grid_sync = nullptr;
  c->impl.big_tmp = nullptr;
  c->impl.big_tmp_bytes = 0;
  c->impl.n_side = 0; c->impl.side_fork = nullptr;
  c->impl.aux_ev[0] = c->impl.aux_ev[1] = nullptr;
  c->impl.defer_unique = 0; c->impl.unique_pending = false; c->impl.unique_ev = nullptr;
  for (int k = 0; k < gdmix_ctx_impl::MAX_SIDE; ++k) { c->impl.side[k] = nullptr; c->impl.side_join[k] = nullptr; }
  for (int k = 0; k < GDMIX_RE_NUM_CLASSES; ++k) { c->impl.ev0[k] = nullptr; c->impl.ev1[k] = nullptr; c->impl.ev_used[k] = false; }
  for (uint32_t& q : c->impl.mail_seq) q = 0;
  hipError_t rc = hipHostMalloc(reinterpret_cast<void**>(&c->impl.host_pinned), HOST_PINNED_BYTES, hipHostMallocMapped | hipHostMallocCoherent);
  if (rc == hipSuccess) memset(c->impl.host_pinned, 0, HOST_PINNED_BYTES);
  if (rc != hipSuccess) {
    set_error("hipHostMalloc failed: %s", hipGetErrorString(rc));
    delete c;
    return GDMIX_RE_EHIP;
  }
  {
    const char* e = getenv("GDMIX_RE_SIDE_STREAM");   // test hook: 0 = every class on the caller's stream; N = that many side streams
    int want = e ? atoi(e) : 3;   // with the caller's stream: the four hardware queues a process gets by default
    if (want > gdmix_ctx_impl::MAX_SIDE) want = gdmix_ctx_impl::MAX_SIDE;
    if (want > 0) {
      rc = hipEventCreateWithFlags(&c->impl.side_fork, hipEventDisableTiming);
      for (int k = 0; k < 2 && rc == hipSuccess; ++k) rc = hipEventCreateWithFlags(&c->impl.aux_ev[k], hipEventDisableTiming);
      if (rc == hipSuccess) rc = hipEventCreateWithFlags(&c->impl.unique_ev, hipEventDisableTiming);
      for (int k = 0; k < want && rc == hipSuccess; ++k) {
        rc = hipStreamCreateWithFlags(&c->impl.side[k], hipStreamNonBlocking);
        if (rc == hipSuccess) rc = hipEventCreateWithFlags(&c->impl.side_join[k], hipEventDisableTiming);
        if (rc == hipSuccess) c->impl.n_side = k + 1;
      }
      if (rc != hipSuccess) {
        set_error("creating the side streams failed: %s", hipGetErrorString(rc));
        gdmix_re_destroy(c);
        return GDMIX_RE_EHIP;
      }
    }
  }
  rc = hipMalloc(&c->impl.grid_sync, TEAM_MAX_TEAMS * sizeof(TeamSync) + TALL_VARIANTS * TALL_TAIL_BYTES + TALL_TEAM_BYTES);
  if (rc != hipSuccess) {
    set_error("hipMalloc failed: %s", hipGetErrorString(rc));
    (void)hipHostFree(c->impl.host_pinned);
    delete c;
    return GDMIX_RE_EHIP;
  }
  *out = c;
  return GDMIX_RE_OK;
}

GDMIX_API void gdmix_re_destroy(gdmix_re_ctx* ctx) {
  if (!ctx) return;
  if (ctx->impl.unique_pending && ctx->impl.unique_ev) (void)hipEventSynchronize(ctx->impl.unique_ev);   // a deferred kernel nobody waited for
  if (ctx->impl.unique_ev) (void)hipEventDestroy(ctx->impl.unique_ev);
  if (ctx->impl.grid_sync) (void)hipFree(ctx->impl.grid_sync);
  if (ctx->impl.big_tmp) (void)hipFree(ctx->impl.big_tmp);
  if (ctx->impl.host_pinned) (void)hipHostFree(ctx->impl.host_pinned);
  if (ctx->impl.side_fork) (void)hipEventDestroy(ctx->impl.side_fork);
  for (int k = 0; k < 2; ++k) if (ctx->impl.aux_ev[k]) (void)hipEventDestroy(ctx->impl.aux_ev[k]);
  for (int k = 0; k < gdmix_ctx_impl::MAX_SIDE; ++k) {
    if (ctx->impl.side_join[k]) (void)hipEventDestroy(ctx->impl.side_join[k]);
    if (ctx->impl.side[k]) (void)hipStreamDestroy(ctx->impl.side[k]);
  }
  for (int k = 0; k < GDMIX_RE_NUM_CLASSES; ++k) {
    if (ctx->impl.ev0[k]) (void)hipEventDestroy(ctx->impl.ev0[k]);
    if (ctx->impl.ev1[k]) (void)hipEventDestroy(ctx->impl.ev1[k]);
  }
  delete ctx;
}

GDMIX_API size_t gdmix_re_pack_workspace_bytes(int64_t E, int64_t N, int64_t Z) {
  if (E < 0 || N < 0 || Z < 0) return 0;
  return pack_workspace_bytes(E, N, Z);
}

GDMIX_API int gdmix_re_pack(gdmix_re_ctx* ctx, const gdmix_re_raw_batch* raw_dev, int has_intercept, void* workspace,
                  size_t workspace_bytes, gdmix_re_packed* out, void* stream) {
  if (!ctx || !raw_dev || !out || (!workspace && workspace_bytes)) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  if (raw_dev->E > 0 && (!raw_dev->ent_row_ptr || !raw_dev->row_nnz_ptr || !raw_dev->y || !raw_dev->offset)) {
    set_error("raw batch has NULL arrays");
    return GDMIX_RE_EINVAL;
  }
  if (raw_dev->Z > 0 && (!raw_dev->col_global || !raw_dev->val)) { set_error("raw batch has NULL arrays"); return GDMIX_RE_EINVAL; }
  HIP_TRY(hipSetDevice(ctx->impl.device));
  HIP_TRY(join_unique(&ctx->impl, static_cast<hipStream_t>(stream)));   // (the workspace of the pack before may be this one)
  return pack_impl(&ctx->impl, raw_dev, has_intercept, workspace, workspace_bytes, out, static_cast<hipStream_t>(stream));
}

GDMIX_API int gdmix_re_set_defer_unique(gdmix_re_ctx* ctx, int enabled) {
  if (!ctx) { set_error("ctx is NULL"); return GDMIX_RE_EINVAL; }
  ctx->impl.defer_unique = (enabled && ctx->impl.n_side > 0) ? 1 : 0;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_pack_join(gdmix_re_ctx* ctx, void* stream) {
  if (!ctx) { set_error("ctx is NULL"); return GDMIX_RE_EINVAL; }
  HIP_TRY(hipSetDevice(ctx->impl.device));
  HIP_TRY(join_unique(&ctx->impl, static_cast<hipStream_t>(stream)));
  return GDMIX_RE_OK;
}

static int slots_for(const gdmix_re_packed* b, const gdmix_re_opts* o, size_t* slot_doubles) {
  *slot_doubles = block_slot_doubles(b->max_p, b->max_n, o->m);
  // enough workgroups to fill the chip a few times over, bounded to ~4 GiB of scratch
  size_t bytes = *slot_doubles * 8;
  size_t by_budget = ((size_t)4 << 30) / (bytes ? bytes : 1);
  size_t slots = 1024;
  if (slots > by_budget) slots = by_budget;
  if (slots < 8) slots = 8;
  if ((int64_t)slots > b->E) slots = (size_t)(b->E > 0 ? b->E : 1);
  return (int)slots;
}

static int64_t var_small_p(const gdmix_re_packed* b) { return b->max_p < VAR_FULL_MAX_P ? b->max_p : VAR_FULL_MAX_P; }

static int var_slots_for(const gdmix_re_packed* b) {
  size_t bytes = var_full_slot_doubles(var_small_p(b)) * 8;
  size_t slots = ((size_t)2 << 30) / (bytes ? bytes : 1);
  if (slots > 2048) slots = 2048;
  if (slots < 4) slots = 4;
  slots &= ~(size_t)3;
  return (int)slots;
}

GDMIX_API size_t gdmix_re_solve_scratch_bytes(const gdmix_re_packed* batch, const gdmix_re_opts* opts) {
  if (!batch || !opts || batch->E == 0) return 0;
  size_t slot_doubles;
  int slots = slots_for(batch, opts, &slot_doubles);
  size_t need = (size_t)slots * slot_doubles * 8;
  if (opts->variance_mode == GDMIX_RE_VAR_FULL && batch->max_p <= VAR_FULL_BIG_MAX_P) {
    size_t v = (size_t)var_slots_for(batch) * var_full_slot_doubles(var_small_p(batch)) * 8;
    if (v > need) need = v;
    if (batch->max_p > VAR_FULL_MAX_P) {
      v = var_full_big_doubles(batch->max_p, batch->max_n) * 8;
      if (v > need) need = v;
    }
  }
  return need;
}

GDMIX_API int gdmix_re_set_wave_lds_limit(gdmix_re_ctx* ctx, int bytes) {
  if (!ctx || bytes < 0) { set_error("bad argument"); return GDMIX_RE_EINVAL; }
  ctx->impl.wave_lds_limit = bytes > 65536 ? 65536 : bytes;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_set_kernel_mask(gdmix_re_ctx* ctx, int mask) {
  if (!ctx) { set_error("ctx is NULL"); return GDMIX_RE_EINVAL; }
  ctx->impl.kernel_mask = mask & 7;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_set_giant_nnz(gdmix_re_ctx* ctx, int64_t nnz) {
  if (!ctx || nnz < 0) { set_error("bad argument"); return GDMIX_RE_EINVAL; }
  ctx->impl.giant_nnz = nnz;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_set_team_nnz(gdmix_re_ctx* ctx, int64_t nnz) {
  if (!ctx || nnz < 0) { set_error("bad argument"); return GDMIX_RE_EINVAL; }
  ctx->impl.team_nnz = nnz;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_set_tall_min_n(gdmix_re_ctx* ctx, int min_n) {
  if (!ctx || min_n < 0) { set_error("bad argument"); return GDMIX_RE_EINVAL; }
  ctx->impl.tall_min_n = min_n;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_set_spread(gdmix_re_ctx* ctx, int queues) {
  if (!ctx || queues < 0) { set_error("bad argument"); return GDMIX_RE_EINVAL; }
  ctx->impl.spread = queues > 1 ? queues : 0;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_set_tall_team_n(gdmix_re_ctx* ctx, int team_n) {
  if (!ctx) { set_error("bad argument"); return GDMIX_RE_EINVAL; }
  ctx->impl.tall_team_n = ctx->impl.num_cus < 8 * TALL_TEAM_C ? 0 : team_n;   // (a device too small for a round of teams never gets the class)
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_device_shared(gdmix_re_ctx* ctx) {
  if (!ctx) { set_error("ctx is NULL"); return GDMIX_RE_EINVAL; }
  return device_has_another_process(ctx->impl.device) ? 1 : 0;
}

GDMIX_API int gdmix_re_set_tall_mid_n(gdmix_re_ctx* ctx, int mid_n) {
  if (!ctx) { set_error("bad argument"); return GDMIX_RE_EINVAL; }
  ctx->impl.tall_mid_n = mid_n < 0 ? -(ctx->impl.num_cus * TALL_MID_WGS) : mid_n;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_set_tall_split_n(gdmix_re_ctx* ctx, int split_n) {
  if (!ctx || split_n < 0) { set_error("bad argument"); return GDMIX_RE_EINVAL; }
  ctx->impl.tall_split_n = split_n > 0 ? split_n : GDMIX_RE_TALL_SPLIT_N_DEFAULT;
  ctx->impl.tall_split_set = split_n > 0 ? 1 : 0;     // 0: back to the default with its per-batch adaptation
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_set_timing(gdmix_re_ctx* ctx, int enabled) {
  if (!ctx) { set_error("ctx is NULL"); return GDMIX_RE_EINVAL; }
  if (enabled && !ctx->impl.ev0[0]) {
    HIP_TRY(hipSetDevice(ctx->impl.device));
    for (int k = 0; k < GDMIX_RE_NUM_CLASSES; ++k) {
      HIP_TRY(hipEventCreate(&ctx->impl.ev0[k]));
      HIP_TRY(hipEventCreate(&ctx->impl.ev1[k]));
    }
  }
  ctx->impl.timing = enabled ? 1 : 0;
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_last_solve_ms(gdmix_re_ctx* ctx, float* ms_out) {
  if (!ctx || !ms_out) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  for (int k = 0; k < GDMIX_RE_NUM_CLASSES; ++k) {
    ms_out[k] = 0.0f;
    if (ctx->impl.ev_used[k]) {
      HIP_TRY(hipEventSynchronize(ctx->impl.ev1[k]));
      HIP_TRY(hipEventElapsedTime(&ms_out[k], ctx->impl.ev0[k], ctx->impl.ev1[k]));
    }
  }
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_set_scratch(gdmix_re_ctx* ctx, void* scratch, size_t bytes) {
  if (!ctx) { set_error("ctx is NULL"); return GDMIX_RE_EINVAL; }
  ctx->impl.scratch = scratch;
  ctx->impl.scratch_bytes = bytes;
  return GDMIX_RE_OK;
}

// diag((X~' D X~ + (l2 + 1e-12) I - l2 e0 e0')^-1) of every entity at `theta` (binary_logistic_regression.py:181-187): one
// wavefront per entity up to p = 2048, the whole device on one entity at a time above (re_variance_big.hip).
static int run_variance_full(gdmix_re_ctx* ctx, const gdmix_re_packed* b, const BatchDev& B, const SolveParams& P, const double* theta,
                             double* variance, hipStream_t s) {
  const size_t vslot = var_full_slot_doubles(var_small_p(b));
  int vslots = var_slots_for(b);
  size_t avail = ctx->impl.scratch ? ctx->impl.scratch_bytes : 0;
  void* base = ctx->impl.scratch;
  if (b->scratch_bytes > avail) { avail = b->scratch_bytes; base = b->scratch; }
  if ((size_t)vslots * vslot * 8 > avail) vslots = (int)(avail / (vslot * 8)) & ~3;
  if (vslots < 4) {
    set_error("variance_mode FULL needs >= %zu bytes of scratch (gdmix_re_set_scratch)", 4 * vslot * 8);
    return GDMIX_RE_ENOMEM;
  }
  HIP_TRY(launch_variance_full(B, b->E, P, theta, variance, static_cast<double*>(base), vslot, vslots, var_small_p(b), s));
  if (b->max_p > VAR_FULL_MAX_P) {   // the large entities one by one, the whole device on each
    if (var_full_big_doubles(b->max_p, b->max_n) * 8 > avail) {
      set_error("variance_mode FULL with p = %d needs >= %zu bytes of scratch (gdmix_re_set_scratch)", b->max_p,
                var_full_big_doubles(b->max_p, b->max_n) * 8);
      return GDMIX_RE_ENOMEM;
    }
    HIP_TRY(launch_variance_full_big(&ctx->impl, B, b->E, P, theta, variance, static_cast<double*>(base), b->max_p, b->max_n, s));
  }
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_solve(gdmix_re_ctx* ctx, const gdmix_re_packed* b, const gdmix_re_opts* opts, const double* theta0,
                   const gdmix_re_result* out, void* stream) {
  if (!ctx || !b || !opts || !out) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  if (opts->m < 1 || opts->max_iter < 0 || opts->maxls < 1) { set_error("bad solver options (m, max_iter, maxls)"); return GDMIX_RE_EINVAL; }
  // regularize_bias without an intercept is legal for the random effect (REParams skips LRParams' check,
  // random_effect_lr_lbfgs_model.py:48-53): the whole theta is regularised (binary_logistic_regression.py:72-82)
  if (opts->variance_mode == GDMIX_RE_VAR_FULL) {
    if (!out->theta) { set_error("variance_mode FULL needs out->theta"); return GDMIX_RE_EINVAL; }
    if (b->max_p > VAR_FULL_BIG_MAX_P) {
      set_error("variance_mode FULL densifies a p x p Hessian per entity; largest entity has p = %d > %lld", b->max_p,
                (long long)VAR_FULL_BIG_MAX_P);
      return GDMIX_RE_ERANGE;
    }
  }
  if (opts->variance_mode != GDMIX_RE_VAR_NONE && !out->variance) { set_error("variance requested but out->variance is NULL"); return GDMIX_RE_EINVAL; }
  if (b->E == 0) return GDMIX_RE_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(hipSetDevice(ctx->impl.device));
  const int ic = opts->has_intercept ? 1 : 0;

  ClassTable tab;
  for (int c = 0; c < GDMIX_RE_NUM_CLASSES; ++c) {
    tab.kind[c] = kClasses[c].kind;
    tab.ncap[c] = kClasses[c].ncap;
    tab.zcap[c] = kClasses[c].zcap;
    int lds = kClasses[c].lds;
    const int gl = group_lanes(kClasses[c].kind);
    if (lds > 0 && gl > 0)
      lds = (gl >= WAVE ? 1 : WAVE / gl) * quad_layout(gl * group_epl(kClasses[c].kind), kClasses[c].ncap, kClasses[c].zcap, gl > WAVE ? gl / WAVE : 1, gl >= WAVE ? 1 : WAVE / gl).bytes;
    bool on = lds > 0 && (lds <= ctx->impl.wave_lds_limit || (gl > WAVE && ctx->impl.wave_lds_limit >= 65536 && lds <= 160 * 1024));
    if (gl > 0 && !(ctx->impl.kernel_mask & 4)) on = false;
    if (kClasses[c].kind == KIND_WLDS && !(ctx->impl.kernel_mask & 2)) on = false;
    tab.lds_bytes[c] = on ? lds : 0;
  }
  // the compact-form team kernels keep TEAM_MCAP history pairs
  tab.giant_nnz = opts->m <= TEAM_MCAP ? ctx->impl.giant_nnz : 0;
  tab.team_nnz = opts->m <= TEAM_MCAP ? ctx->impl.team_nnz : 0;
  tab.tall_min_n = ctx->impl.tall_min_n;
  tab.tall_split_n = ctx->impl.tall_split_n;
  tab.tall_adapt_limit = ctx->impl.tall_split_set ? 0 : ctx->impl.tall_adapt_limit;   // an explicit split is kept, whatever its value
  {   // > 0: adaptive from team_n on; < 0: everything from -team_n on (tests); never below TALL_TEAM_MIN_N samples
    const int tn = ctx->impl.tall_team_n;
    tab.tall_team_n = tn > 0 ? tn : -tn;
    if (tab.tall_team_n > 0 && tab.tall_team_n < TALL_TEAM_MIN_N) tab.tall_team_n = TALL_TEAM_MIN_N;
    tab.tall_team_limit = tn > 0 ? ctx->impl.tall_team_limit : 0;
  }
  // the mid class adapts with the split: a caller who pinned the split (gdmix_re_set_tall_split_n) pinned the routing — no per-batch class
  tab.tall_mid_n = (ctx->impl.tall_mid_n < 0 && ctx->impl.tall_split_set) ? 0 : ctx->impl.tall_mid_n;
  if (opts->sum_loss || opts->linear) {
    // the fixed-effect objective lives in the team kernels only: every entity goes device-wide, one after another
    if (opts->m > TEAM_MCAP) { set_error("sum_loss / linear need m <= %d", TEAM_MCAP); return GDMIX_RE_EINVAL; }
    if (opts->variance_mode != GDMIX_RE_VAR_NONE) { set_error("variance is not available with sum_loss / linear"); return GDMIX_RE_EINVAL; }
    tab.giant_nnz = 1;
  }

  int32_t* cc = b->class_count;
  HIP_TRY(hipMemsetAsync(cc, 0, 6 * GDMIX_RE_NUM_CLASSES * sizeof(int32_t), s));
  HIP_TRY(launch_classify(b, ic, opts->m, tab, b->cls_tmp, cc, s));
  hipLaunchKernelGGL(class_base_kernel, dim3(1), dim3(1), 0, s, cc, tab.tall_adapt_limit, 16 * ctx->impl.num_cus, tab.tall_team_n, tab.tall_team_limit, tab.tall_mid_n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(launch_order(b, b->cls_tmp, cc + GDMIX_RE_NUM_CLASSES, cc + 2 * GDMIX_RE_NUM_CLASSES, s));
  int32_t* hc = ctx->impl.host_pinned + 256;
  HIP_TRY(fetch_small(&ctx->impl, 0, cc, 6 * GDMIX_RE_NUM_CLASSES * sizeof(int32_t), hc, s));

  SolveParams P;
  P.l2 = opts->l2; P.ftol = opts->ftol; P.pgtol = opts->pgtol; P.threshold = opts->threshold;
  P.regularize_bias = opts->regularize_bias; P.has_intercept = ic; P.m = opts->m; P.max_iter = opts->max_iter;
  P.maxfun = opts->maxfun; P.maxls = opts->maxls; P.variance_mode = opts->variance_mode;
  P.sum_loss = opts->sum_loss ? 1 : 0; P.linear = opts->linear ? 1 : 0;
  BatchDev B = make_batch_dev(b);
  OutDev O{out->theta, out->theta_thr, out->variance, out->fval, out->gnorm, out->nit, out->nfev, out->status};

  const bool timing = ctx->impl.timing != 0;
  for (int c = 0; c < GDMIX_RE_NUM_CLASSES; ++c) ctx->impl.ev_used[c] = false;
  // scratch slots of the workgroup-per-entity and team kernels
  size_t slot_doubles = 0;
  int slots = 0;
  double* slot_scratch = nullptr;
  const int n_slot_users = hc[BLOCK_CLASS] + hc[TEAM128_CLASS] + hc[TEAM32_CLASS] + hc[TEAM8_CLASS] + hc[GIANT_CLASS];
  if (n_slot_users > 0) {
    slots = slots_for(b, opts, &slot_doubles);
    size_t need = (size_t)slots * slot_doubles * 8;
    if (ctx->impl.scratch && ctx->impl.scratch_bytes >= need) slot_scratch = static_cast<double*>(ctx->impl.scratch);
    else if (b->scratch && b->scratch_bytes >= need) slot_scratch = static_cast<double*>(b->scratch);
    else {
      // shrink the slot count to what is available rather than failing, but never below one slot
      size_t avail = ctx->impl.scratch ? ctx->impl.scratch_bytes : 0;
      void* base = ctx->impl.scratch;
      if (b->scratch_bytes > avail) { avail = b->scratch_bytes; base = b->scratch; }
      slots = (int)(avail / (slot_doubles * 8));
      if (slots < 1) {
        set_error("%d entities need a scratch slot: provide >= %zu bytes via gdmix_re_set_scratch", n_slot_users, slot_doubles * 8);
        return GDMIX_RE_ENOMEM;
      }
      slot_scratch = static_cast<double*>(base);
    }
  }
  // A lean tall class of a few thousand entities is not worth a launch of its own (a launch lasts at least one entity's solve and
  // ends in a thin tail): it then runs with the class behind it — its entities sit right in front of that class's in `order`, and
  // the general one-wavefront kernel takes any of them. An entity's result does not depend on which of the two ran it: same
  // accumulator sets, same order of the adds (the variants differ in where loads are issued and in occupancy only).
  int lean_rounds = 4;   // (GDMIX_RE_LEAN_MERGE_ROUNDS: exploration knob)
  if (const char* ev = getenv("GDMIX_RE_LEAN_MERGE_ROUNDS")) lean_rounds = atoi(ev);
  const int lean_merged = (hc[TALL_L_CLASS] > 0 && hc[TALL_L_CLASS] < lean_rounds * ctx->impl.num_cus * TALL_LEAN_WGS) ? hc[TALL_L_CLASS] : 0;
  int n_launch_classes = 0;
  for (int c = 0; c < GDMIX_RE_NUM_CLASSES; ++c) n_launch_classes += (hc[c] > 0 && !(c == TALL_L_CLASS && lean_merged)) ? 1 : 0;
  hipStream_t const s_main = s;
  // the side stream starts where the caller's stream stands now (the class lists are ready), not where it stands when the first
  // small class comes up in the loop: that one may be the last
  // Once the side stream has forked, EVERY way out of this function joins it back into the caller's stream (an error
  // between fork and join used to return with side-stream kernels still running on buffers the caller may then free or reuse).
  SideJoin side_join{&ctx->impl, s_main};
  bool forked = false;
  if (ctx->impl.n_side > 0 && n_launch_classes > 1) {
    for (int c = 0; c < BLOCK_CLASS && !forked; ++c) {
      int cnt = hc[c];
      if (c == TALL_L_CLASS && lean_merged) continue;
      if (c == TALL_S_CLASS) cnt += lean_merged;
      forked = cnt > 0 && class_is_small(kClasses[c].kind, cnt, ctx->impl.num_cus);
    }
    if (ctx->impl.spread > 1) forked = true;
    if (forked) HIP_TRY(hipEventRecord(ctx->impl.side_fork, s_main));
  }
  // Launch plan. The tall classes go first: their kernels are the longest chains of a small batch (one workgroup, or one wavefront,
  // per entity, for as long as that entity's solve lasts) and each is preceded by three small launches (fill, sort, tail) that must
  // not queue behind the group kernels' workgroups. Stream of a class:
  //   large class                      -> dealt over the caller's stream and the side streams in launch order (gdmix_re_set_spread);
  //   small tall class (<8>, <1>, lean) -> a side stream of its own (side 1, 2, 0: the device has four hardware queues by default,
  //                                        a fourth side stream would share one — measured: both tall classes of a MovieLens share on
  //                                        one queue, 1.1 + 2.1 ms one after the other);
  //   other small classes              -> the caller's stream when no large class uses it, else side 0 (behind the lean tall class,
  //                                        the shortest of the three).
  // Only the streams that get work are woken (SideJoin::use): a C2 partition touches the caller's stream and side 0.
  // Disjoint entities and outputs; the tall variants have a tail slot each: a schedule changes the time, never a bit of the result.
  struct Launch { int c, b0, cnt; bool small; int spread; };
  Launch plan[GDMIX_RE_NUM_CLASSES];
  int n_plan = 0;
  bool any_large = false;
  {
    int begin = 0;
    int b0s[GDMIX_RE_NUM_CLASSES], cnts[GDMIX_RE_NUM_CLASSES];
    for (int c = 0; c < BLOCK_CLASS; ++c) {
      b0s[c] = begin; cnts[c] = hc[c];
      begin += hc[c];
      if (c == TALL_L_CLASS && lean_merged) cnts[c] = 0;
      if (c == TALL_S_CLASS && lean_merged) { b0s[c] -= lean_merged; cnts[c] += lean_merged; }
    }
    const int first[5] = {TALL_T_CLASS, TALL_CLASS, TALL_M_CLASS, TALL_S_CLASS, TALL_L_CLASS};
    for (int k = 0; k < 5 + BLOCK_CLASS; ++k) {
      const int c = k < 5 ? first[k] : k - 5;
      if (k >= 5 && (c == TALL_T_CLASS || c == TALL_CLASS || c == TALL_M_CLASS || c == TALL_S_CLASS || c == TALL_L_CLASS)) continue;
      if (cnts[c] <= 0) continue;
      const bool small = forked && class_is_small(kClasses[c].kind, cnts[c], ctx->impl.num_cus);
      any_large = any_large || !small;
      plan[n_plan++] = Launch{c, b0s[c], cnts[c], small, -1};
    }
  }
  int begin = 0;
  int spread_rr = 0;
  for (int c = 0; c < BLOCK_CLASS; ++c) begin += hc[c];
  for (int k = 0; k < n_plan; ++k) {
    const int c = plan[k].c, b0 = plan[k].b0, cnt = plan[k].cnt;
    hipStream_t s = s_main;
    if (!plan[k].small && forked && ctx->impl.spread > 1) {   // large classes side by side
      const int nq = ctx->impl.spread < ctx->impl.n_side + 1 ? ctx->impl.spread : ctx->impl.n_side + 1;
      const int q = spread_rr++ % nq;
      if (q > 0) { HIP_TRY(side_join.use(q - 1)); s = ctx->impl.side[q - 1]; }
    }
    if (plan[k].small) {
      const int ns = ctx->impl.n_side;
      int k_side = -1;
      if (c == TALL_CLASS) k_side = 1 % ns;
      else if (c == TALL_S_CLASS) k_side = 2 % ns;
      else if (c == TALL_M_CLASS && !any_large) k_side = -1;      // the caller's stream, ahead of the group classes (three side streams: a fourth would share a hardware queue)
      else if (c == TALL_T_CLASS || c == TALL_L_CLASS || any_large) k_side = 0;
      if (k_side >= 0) { HIP_TRY(side_join.use(k_side)); s = ctx->impl.side[k_side]; }
    }
    if (timing) { HIP_TRY(hipEventRecord(ctx->impl.ev0[c], s)); }
    switch (kClasses[c].kind) {
      case KIND_QUAD2: case KIND_QUAD3: case KIND_QUAD4: case KIND_PAIR3: case KIND_PAIR4:
      case KIND_G64_3: case KIND_G64_4: case KIND_G128_3: case KIND_G128_4: case KIND_G256_3: case KIND_G256_4: case KIND_G512_4:
        HIP_TRY(launch_solve_quad(group_lanes(kClasses[c].kind), group_epl(kClasses[c].kind), B, O, P, theta0, b0, cnt,
                                  kClasses[c].ncap, kClasses[c].zcap, s));
        break;
      case KIND_TALL_T: {
        const char* xe = getenv("GDMIX_RE_XCD_BARRIER");   // 0: every signal with the full release (A/B and tests)
        HIP_TRY(launch_solve_tall_team(B, O, P, theta0, b0, cnt, ctx->impl.num_cus, b->Z, static_cast<char*>(ctx->impl.grid_sync) + TEAM_MAX_TEAMS * sizeof(TeamSync) + TALL_VARIANT_TEAM * TALL_TAIL_BYTES,
                                       static_cast<char*>(ctx->impl.grid_sync) + TEAM_MAX_TEAMS * sizeof(TeamSync) + TALL_VARIANTS * TALL_TAIL_BYTES, (xe && atoi(xe) == 0) ? 0 : 1, s));
        break;
      }
      case KIND_TALL_L: HIP_TRY(launch_solve_tall(TALL_VARIANT_LEAN, B, O, P, theta0, b0, cnt, ctx->impl.num_cus, b->Z, static_cast<char*>(ctx->impl.grid_sync) + TEAM_MAX_TEAMS * sizeof(TeamSync) + TALL_VARIANT_LEAN * TALL_TAIL_BYTES, static_cast<TeamSync*>(ctx->impl.grid_sync) + TALL_VARIANT_LEAN, 0, s)); break;
      case KIND_TALL_S: HIP_TRY(launch_solve_tall(TALL_VARIANT_SMALL, B, O, P, theta0, b0, cnt, ctx->impl.num_cus, b->Z, static_cast<char*>(ctx->impl.grid_sync) + TEAM_MAX_TEAMS * sizeof(TeamSync) + TALL_VARIANT_SMALL * TALL_TAIL_BYTES, static_cast<TeamSync*>(ctx->impl.grid_sync) + TALL_VARIANT_SMALL, lean_merged, s)); break;
      case KIND_TALL_M: HIP_TRY(launch_solve_tall(TALL_VARIANT_MID, B, O, P, theta0, b0, cnt, ctx->impl.num_cus, b->Z, static_cast<char*>(ctx->impl.grid_sync) + TEAM_MAX_TEAMS * sizeof(TeamSync) + TALL_VARIANT_MID * TALL_TAIL_BYTES, static_cast<TeamSync*>(ctx->impl.grid_sync) + TALL_VARIANT_MID, 0, s)); break;
      case KIND_TALL: HIP_TRY(launch_solve_tall(TALL_VARIANT_LARGE, B, O, P, theta0, b0, cnt, ctx->impl.num_cus, b->Z, static_cast<char*>(ctx->impl.grid_sync) + TEAM_MAX_TEAMS * sizeof(TeamSync) + TALL_VARIANT_LARGE * TALL_TAIL_BYTES, static_cast<TeamSync*>(ctx->impl.grid_sync) + TALL_VARIANT_LARGE, 0, s)); break;
      default: HIP_TRY(launch_solve_wave(B, O, P, theta0, b0, cnt, kClasses[c].lds, s)); break;
    }
    if (timing) { HIP_TRY(hipEventRecord(ctx->impl.ev1[c], s)); ctx->impl.ev_used[c] = true; }
  }
  side_join.join();   // before the team kernels (they use the whole device and the scratch slots)
  if (hc[BLOCK_CLASS] > 0 || hc[TEAM128_CLASS] > 0 || hc[TEAM32_CLASS] > 0 || hc[TEAM8_CLASS] > 0 || hc[GIANT_CLASS] > 0) {
    double* const scratch = slot_scratch;
    if (hc[BLOCK_CLASS] > 0) {
      if (timing) { HIP_TRY(hipEventRecord(ctx->impl.ev0[BLOCK_CLASS], s)); }
      HIP_TRY(launch_solve_block(B, O, P, theta0, begin, hc[BLOCK_CLASS], scratch, slot_doubles, slots, b->max_p, s));
      if (timing) { HIP_TRY(hipEventRecord(ctx->impl.ev1[BLOCK_CLASS], s)); ctx->impl.ev_used[BLOCK_CLASS] = true; }
      begin += hc[BLOCK_CLASS];
    }
    // the team kernels run after the workgroup kernel on the same stream, so its slots are free again
    for (int tier = 0; tier < 3; ++tier) {
      const int cls = tier == 0 ? TEAM128_CLASS : (tier == 1 ? TEAM32_CLASS : TEAM8_CLASS);
      if (hc[cls] <= 0) continue;
      // Team count: the longest solve on its team should take about as long as the whole tier on the device, i.e.
      // teams ~ total work / largest entity's work (non-zeros as the measure), within what the tier allows: few large
      // entities get large teams (the critical path matters), many small ones small teams (the barriers do).
      const int tier_max = tier == 0 ? 128 : (tier == 1 ? 32 : 8);
      const unsigned long long tot = reinterpret_cast<const unsigned long long*>(hc + 4 * GDMIX_RE_NUM_CLASSES)[cls];
      const unsigned long long big = (unsigned long long)hc[3 * GDMIX_RE_NUM_CLASSES + cls];
      // teams ~ total work / largest entity's work, as a power of two: rounded down when the tier has many entities (the
      // barriers of many small teams are the cost); a tier of a handful of entities gets about one team per entity (5
      // entities of 0.2 .. 0.9 M non-zeros: 9.4 ms on one team of 256 CUs, 6.5 ms on four of 64; a makespan model with
      // fixed per-CU constants did worse than these two rules on the larger tiers)
      const unsigned long long ratio = big ? tot / big : 1;
      int teams = 1;
      while (teams * 2 <= tier_max && (unsigned long long)(teams * 2) <= ratio) teams *= 2;
      if (hc[cls] <= 16) {   // a handful: one team per entity, as far as a power of two, the tier and 4x the rule above allow
        // (powers of two keep a team on few XCDs: 6 entities on 6 teams of 42 CUs, each spread over all eight XCDs, took 39 ms
        // against 28 ms on 4 teams of 64)
        int want = 1;
        while (want * 2 <= hc[cls] && want * 2 <= tier_max && want * 2 <= teams * 4) want *= 2;
        if (want > teams) teams = want;
      }
      if (const char* ev = getenv("GDMIX_RE_TEAMS")) teams = atoi(ev);   // exploration knob
      if (teams > TEAM_MAX_TEAMS) teams = TEAM_MAX_TEAMS;
      while (teams > 1 && (slots < teams || ctx->impl.num_cus % teams)) teams >>= 1;
      if (teams < 1) teams = 1;
      if (timing) { HIP_TRY(hipEventRecord(ctx->impl.ev0[cls], s)); }
      HIP_TRY(launch_solve_grid(B, O, P, theta0, begin, hc[cls], scratch, slot_doubles, b->max_p, ctx->impl.grid_sync,
                                ctx->impl.num_cus, teams, s));
      if (timing) { HIP_TRY(hipEventRecord(ctx->impl.ev1[cls], s)); ctx->impl.ev_used[cls] = true; }
      begin += hc[cls];
    }
    if (hc[GIANT_CLASS] > 0) {
      if (timing) { HIP_TRY(hipEventRecord(ctx->impl.ev0[GIANT_CLASS], s)); }
      HIP_TRY(launch_solve_grid(B, O, P, theta0, begin, hc[GIANT_CLASS], scratch, slot_doubles, b->max_p,
                                ctx->impl.grid_sync, ctx->impl.num_cus, 1, s));
      if (timing) { HIP_TRY(hipEventRecord(ctx->impl.ev1[GIANT_CLASS], s)); ctx->impl.ev_used[GIANT_CLASS] = true; }
    }
  }
  if (opts->variance_mode == GDMIX_RE_VAR_FULL) {
    const int rc = run_variance_full(ctx, b, B, P, out->theta, out->variance, s);
    if (rc != GDMIX_RE_OK) return rc;
  }
  // a pack's deferred compaction ran next to this solve: from here on the caller's stream is behind it, as it is behind the solve
  HIP_TRY(join_unique(&ctx->impl, s_main));
  return GDMIX_RE_OK;
}

GDMIX_API int gdmix_re_variance_full(gdmix_re_ctx* ctx, const gdmix_re_packed* b, const gdmix_re_opts* opts, const double* theta,
                                     double* variance, void* stream) {
  if (!ctx || !b || !opts || !theta || !variance) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  if (b->max_p > VAR_FULL_BIG_MAX_P) {
    set_error("variance_mode FULL densifies a p x p Hessian per entity; largest entity has p = %d > %lld", b->max_p, (long long)VAR_FULL_BIG_MAX_P);
    return GDMIX_RE_ERANGE;
  }
  if (b->E == 0) return GDMIX_RE_OK;
  HIP_TRY(hipSetDevice(ctx->impl.device));
  HIP_TRY(join_unique(&ctx->impl, static_cast<hipStream_t>(stream)));
  SolveParams P;
  P.l2 = opts->l2; P.ftol = opts->ftol; P.pgtol = opts->pgtol; P.threshold = opts->threshold;
  P.regularize_bias = opts->regularize_bias; P.has_intercept = opts->has_intercept ? 1 : 0; P.m = opts->m; P.max_iter = opts->max_iter;
  P.maxfun = opts->maxfun; P.maxls = opts->maxls; P.variance_mode = GDMIX_RE_VAR_FULL;
  P.sum_loss = 0; P.linear = 0;
  return run_variance_full(ctx, b, make_batch_dev(b), P, theta, variance, static_cast<hipStream_t>(stream));
}

GDMIX_API int gdmix_re_score(gdmix_re_ctx* ctx, const gdmix_re_packed* b, int has_intercept, const double* theta,
                   const uint8_t* has_model, float* logit, float* logit_per_coord, void* stream) {
  if (!ctx || !b || !logit || !logit_per_coord) { set_error("NULL argument"); return GDMIX_RE_EINVAL; }
  if (!theta && !has_model) { set_error("theta is NULL"); return GDMIX_RE_EINVAL; }
  HIP_TRY(hipSetDevice(ctx->impl.device));
  HIP_TRY(join_unique(&ctx->impl, static_cast<hipStream_t>(stream)));
  BatchDev B = make_batch_dev(b);
  HIP_TRY(launch_score(B, b->E, b->N, has_intercept ? 1 : 0, theta, has_model, logit, logit_per_coord,
                       static_cast<hipStream_t>(stream)));
  return GDMIX_RE_OK;
}

GDMIX_API const char* gdmix_re_class_kernel_name(int c) {
  if (c < 0 || c >= GDMIX_RE_NUM_CLASSES) return nullptr;
  return kClasses[c].name;
}

GDMIX_API int32_t gdmix_java_string_hash(const uint16_t* utf16, int64_t len) {
  uint32_t h = 0;
  for (int64_t i = 0; i < len; ++i) h = 31u * h + (uint32_t)utf16[i];
  return (int32_t)h;
}

GDMIX_API int32_t gdmix_java_partition_id(const uint16_t* utf16, int64_t len, int32_t num_partitions) {
  if (num_partitions <= 0) return -1;
  const int32_t h = gdmix_java_string_hash(utf16, len);
  const int32_t a = (h == INT32_MIN) ? h : (h < 0 ? -h : h);
  return (int32_t)((int64_t)a % (int64_t)num_partitions);
}

GDMIX_API int gdmix_java_partition_ids_i64(gdmix_re_ctx* ctx, const int64_t* ids_dev, int64_t count, int32_t num_partitions,
                                 int32_t* out_dev, void* stream) {
  if (!ctx || (count > 0 && (!ids_dev || !out_dev)) || num_partitions <= 0) { set_error("bad argument"); return GDMIX_RE_EINVAL; }
  HIP_TRY(hipSetDevice(ctx->impl.device));
  HIP_TRY(launch_partition_ids(ids_dev, count, num_partitions, out_dev, static_cast<hipStream_t>(stream)));
  return GDMIX_RE_OK;
}

}  // extern "C"
