"""ctypes binding of libgdmix_re.so (include/gdmix_re.h) — the device solve behind the model API.

PyTorch-ROCm is used for plumbing only: device buffers (torch tensors), the current HIP stream and
host<->device copies. Every numeric step runs in the hand-written HIP kernels of gdmix_amd/csrc.

There is NO CPU fallback: if the library is missing or no gfx950 device is visible, construction
raises. (The CPU restatement under oracle/ is test infrastructure and is never imported here.)
"""
import ctypes as C
import functools
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .batch import RawBatch, WireRawBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgdmix_re.so")

NUM_CLASSES = 40
STATUS_NAMES = ("PGTOL", "FACTR", "MAXITER", "MAXFUN", "ABNORMAL")
VAR_NONE, VAR_SIMPLE, VAR_FULL = 0, 1, 2
VARIANCE_MODES = {None: VAR_NONE, "simple": VAR_SIMPLE, "SIMPLE": VAR_SIMPLE, "full": VAR_FULL, "FULL": VAR_FULL,
                  0: VAR_NONE, 1: VAR_SIMPLE, 2: VAR_FULL}


class GdmixReError(RuntimeError):
    pass


class _RawBatch(C.Structure):
    _fields_ = [("E", C.c_int64), ("N", C.c_int64), ("Z", C.c_int64),
                ("ent_row_ptr", C.c_void_p), ("row_nnz_ptr", C.c_void_p), ("col_global", C.c_void_p),
                ("val", C.c_void_p), ("y", C.c_void_p), ("offset", C.c_void_p), ("weight", C.c_void_p)]


class _WireBatch(C.Structure):
    _fields_ = [("E", C.c_int64), ("N", C.c_int64), ("Z", C.c_int64), ("ent_n", C.c_void_p), ("row_nnz", C.c_void_p),
                ("row_nnz_width", C.c_int32), ("y_width", C.c_int32), ("col_width", C.c_int32), ("reserved", C.c_int32),
                ("col_global", C.c_void_p), ("val", C.c_void_p),
                ("y", C.c_void_p), ("offset", C.c_void_p), ("weight", C.c_void_p)]


class _Packed(C.Structure):
    _fields_ = [("E", C.c_int64), ("N", C.c_int64), ("Z", C.c_int64), ("D", C.c_int64),
                ("ent_row_ptr", C.c_void_p), ("ent_nnz_ptr", C.c_void_p), ("ent_feat_ptr", C.c_void_p),
                ("row_ptr", C.c_void_p), ("csr_col", C.c_void_p), ("csr_val", C.c_void_p),
                ("col_ptr", C.c_void_p), ("csc_row", C.c_void_p), ("csc_val", C.c_void_p),
                ("unique_global", C.c_void_p), ("y", C.c_void_p), ("offset", C.c_void_p), ("weight", C.c_void_p),
                ("order", C.c_void_p), ("cls_tmp", C.c_void_p), ("class_count", C.c_void_p),
                ("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t),
                ("max_p", C.c_int32), ("max_n", C.c_int32), ("max_nnz", C.c_int32)]


class _Opts(C.Structure):
    _fields_ = [("l2", C.c_double), ("regularize_bias", C.c_int32), ("has_intercept", C.c_int32),
                ("m", C.c_int32), ("max_iter", C.c_int32), ("maxfun", C.c_int32), ("maxls", C.c_int32),
                ("ftol", C.c_double), ("pgtol", C.c_double), ("variance_mode", C.c_int32),
                ("threshold", C.c_double), ("sum_loss", C.c_int32), ("linear", C.c_int32)]


class _Result(C.Structure):
    _fields_ = [("theta", C.c_void_p), ("theta_thr", C.c_void_p), ("variance", C.c_void_p),
                ("fval", C.c_void_p), ("gnorm", C.c_void_p), ("nit", C.c_void_p), ("nfev", C.c_void_p),
                ("status", C.c_void_p)]


EXPORTED_SYMBOLS = (
    "gdmix_re_abi_version", "gdmix_re_build_id", "gdmix_re_device_shared", "gdmix_re_grid_lock_acquire", "gdmix_re_grid_lock_release", "gdmix_re_grid_lock_stats", "gdmix_re_last_error", "gdmix_re_default_opts", "gdmix_re_create",
    "gdmix_re_destroy", "gdmix_re_pack_workspace_bytes", "gdmix_re_pack", "gdmix_re_set_defer_unique", "gdmix_re_pack_join", "gdmix_re_solve",
    "gdmix_re_solve_scratch_bytes", "gdmix_re_set_scratch", "gdmix_re_variance_full", "gdmix_re_set_wave_lds_limit", "gdmix_re_score",
    "gdmix_re_widen_workspace_bytes", "gdmix_re_widen", "gdmix_re_set_timing", "gdmix_re_last_solve_ms", "gdmix_re_set_kernel_mask", "gdmix_re_set_giant_nnz", "gdmix_re_set_team_nnz", "gdmix_re_set_tall_min_n", "gdmix_re_set_tall_split_n", "gdmix_re_set_tall_team_n", "gdmix_re_set_tall_mid_n", "gdmix_re_set_spread",
    "gdmix_fe_create", "gdmix_fe_destroy", "gdmix_fe_eval", "gdmix_fe_reduce_buffer", "gdmix_fe_step", "gdmix_fe_step_async", "gdmix_fe_step_status", "gdmix_fe_solve", "gdmix_fe_result",
    "gdmix_fe_last_eval_ms", "gdmix_fe_stream_bytes", "gdmix_fe_score", "gdmix_fe_hessian_diag", "gdmix_fe_hessian_dense_scratch_bytes", "gdmix_fe_hessian_dense",
    "gdmix_fe_variance_of_hessian",
    "gdmix_re_class_kernel_name", "gdmix_java_string_hash", "gdmix_java_partition_id",
    "gdmix_java_partition_ids_i64")

_lib = None


def load_library():
    """dlopen libgdmix_re.so (built in-tree by gdmix_amd.build). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GdmixReError(
            f"{LIB_PATH} is missing: build it with `python -m gdmix_amd.build` "
            "(there is no CPU fallback for the random-effect solve)")
    try:  # make sure the HIP runtime torch already loaded (if any) is the one the library binds to
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is always present in this image
        pass
    from . import build as _build
    stale = _build.check_library(LIB_PATH, _build.source_id, "csrc/*.hip, *.hpp")
    if stale:   # a prebuilt library that is not the build of the sources it travels with: every number it produced would be mislabelled
        raise GdmixReError(stale)
    lib = C.CDLL(LIB_PATH)
    lib.gdmix_re_abi_version.restype = C.c_int
    lib.gdmix_re_build_id.restype = C.c_char_p
    lib.gdmix_re_device_shared.argtypes = [C.c_void_p]
    lib.gdmix_re_grid_lock_acquire.argtypes = [C.c_char_p]
    lib.gdmix_re_grid_lock_release.argtypes = [C.c_char_p]
    lib.gdmix_re_grid_lock_stats.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.gdmix_re_last_error.restype = C.c_char_p
    lib.gdmix_re_default_opts.argtypes = [C.POINTER(_Opts)]
    lib.gdmix_re_default_opts.restype = None
    lib.gdmix_re_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.gdmix_re_destroy.argtypes = [C.c_void_p]
    lib.gdmix_re_destroy.restype = None
    lib.gdmix_re_pack_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    lib.gdmix_re_pack_workspace_bytes.restype = C.c_size_t
    lib.gdmix_re_pack.argtypes = [C.c_void_p, C.POINTER(_RawBatch), C.c_int, C.c_void_p, C.c_size_t,
                                  C.POINTER(_Packed), C.c_void_p]
    lib.gdmix_re_set_defer_unique.argtypes = [C.c_void_p, C.c_int]
    lib.gdmix_re_pack_join.argtypes = [C.c_void_p, C.c_void_p]
    lib.gdmix_re_widen_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    lib.gdmix_re_widen_workspace_bytes.restype = C.c_size_t
    lib.gdmix_re_widen.argtypes = [C.c_void_p, C.POINTER(_WireBatch), C.c_void_p, C.c_size_t, C.POINTER(_RawBatch), C.c_void_p]
    lib.gdmix_re_solve.argtypes = [C.c_void_p, C.POINTER(_Packed), C.POINTER(_Opts), C.c_void_p,
                                   C.POINTER(_Result), C.c_void_p]
    lib.gdmix_re_variance_full.argtypes = [C.c_void_p, C.POINTER(_Packed), C.POINTER(_Opts), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gdmix_re_solve_scratch_bytes.argtypes = [C.POINTER(_Packed), C.POINTER(_Opts)]
    lib.gdmix_re_solve_scratch_bytes.restype = C.c_size_t
    lib.gdmix_re_set_scratch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.gdmix_re_set_wave_lds_limit.argtypes = [C.c_void_p, C.c_int]
    lib.gdmix_re_set_kernel_mask.argtypes = [C.c_void_p, C.c_int]
    lib.gdmix_re_set_giant_nnz.argtypes = [C.c_void_p, C.c_int64]
    lib.gdmix_fe_create.argtypes = [C.c_void_p, C.POINTER(_Packed), C.c_int64, C.POINTER(_Opts), C.c_void_p,
                                    C.POINTER(C.c_void_p), C.c_void_p]
    lib.gdmix_fe_destroy.argtypes = [C.c_void_p]
    lib.gdmix_fe_destroy.restype = None
    lib.gdmix_fe_eval.argtypes = [C.c_void_p, C.c_void_p]
    lib.gdmix_fe_reduce_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.gdmix_fe_reduce_buffer.restype = C.c_void_p
    lib.gdmix_fe_step.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    lib.gdmix_fe_step_async.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    lib.gdmix_fe_step_status.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]
    lib.gdmix_fe_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.gdmix_fe_hessian_diag.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gdmix_fe_stream_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.gdmix_fe_result.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]
    lib.gdmix_fe_score.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gdmix_fe_hessian_dense_scratch_bytes.argtypes = [C.POINTER(_Packed)]
    lib.gdmix_fe_hessian_dense_scratch_bytes.restype = C.c_size_t
    lib.gdmix_fe_hessian_dense.argtypes = [C.c_void_p, C.POINTER(_Packed), C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.gdmix_fe_variance_of_hessian.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gdmix_fe_last_eval_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.gdmix_re_set_team_nnz.argtypes = [C.c_void_p, C.c_int64]
    lib.gdmix_re_set_tall_min_n.argtypes = [C.c_void_p, C.c_int]
    lib.gdmix_re_set_tall_split_n.argtypes = [C.c_void_p, C.c_int]
    lib.gdmix_re_set_tall_team_n.argtypes = [C.c_void_p, C.c_int]
    lib.gdmix_re_set_tall_mid_n.argtypes = [C.c_void_p, C.c_int]
    lib.gdmix_re_set_spread.argtypes = [C.c_void_p, C.c_int]
    lib.gdmix_re_set_timing.argtypes = [C.c_void_p, C.c_int]
    lib.gdmix_re_last_solve_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.gdmix_re_score.argtypes = [C.c_void_p, C.POINTER(_Packed), C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gdmix_re_class_kernel_name.argtypes = [C.c_int]
    lib.gdmix_re_class_kernel_name.restype = C.c_char_p
    lib.gdmix_java_string_hash.argtypes = [C.c_void_p, C.c_int64]
    lib.gdmix_java_string_hash.restype = C.c_int32
    lib.gdmix_java_partition_id.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    lib.gdmix_java_partition_id.restype = C.c_int32
    lib.gdmix_java_partition_ids_i64.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    if lib.gdmix_re_abi_version() != 12:
        raise GdmixReError("libgdmix_re.so ABI version mismatch")
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().gdmix_re_last_error().decode("utf-8", "replace")
        raise GdmixReError(f"{what} failed ({rc}): {msg}")


@dataclass
class SolverOptions:
    """Solver knobs; defaults are REParams/LRParams defaults (base_lr_params.py:22-27) plus scipy's
    fmin_l_bfgs_b defaults for the arguments the reference leaves unset (pgtol, maxfun, maxls)."""
    l2: float = 1.0
    regularize_bias: bool = True
    has_intercept: bool = True
    m: int = 10
    max_iter: int = 100
    maxfun: int = 15000
    maxls: int = 20
    ftol: float = 1e-12           # = factr*eps = lbfgs_tolerance (random_effect_lr_lbfgs_model.py:142-146)
    pgtol: float = 1e-5
    variance_mode: int = VAR_NONE
    threshold: float = 1e-4       # sparsity_threshold (base_lr_params.py:32)
    sum_loss: bool = False        # fixed-effect objective: not divided by n (fixed_effect_lr_lbfgs_model.py:363-381)
    linear: bool = False          # squared loss instead of the logistic loss (:356-358)

    def to_c(self):
        return _Opts(float(self.l2), int(bool(self.regularize_bias)), int(bool(self.has_intercept)),
                     int(self.m), int(self.max_iter), int(self.maxfun), int(self.maxls), float(self.ftol),
                     float(self.pgtol), int(VARIANCE_MODES[self.variance_mode]), float(self.threshold),
                     int(bool(self.sum_loss)), int(bool(self.linear)))


def java_string_hash(s: str) -> int:
    """String.hashCode of the JVM (PartitionUtils.scala:31-37), over UTF-16 code units."""
    cu = np.frombuffer(s.encode("utf-16-le"), dtype=np.uint16).copy()
    return int(load_library().gdmix_java_string_hash(cu.ctypes.data_as(C.c_void_p), cu.size))


def java_partition_id(s: str, num_partitions: int) -> int:
    """Math.abs(s.hashCode) % numPartitions, bit-exact incl. Int.MinValue (PartitionUtils.scala:31-37)."""
    cu = np.frombuffer(s.encode("utf-16-le"), dtype=np.uint16).copy()
    return int(load_library().gdmix_java_partition_id(cu.ctypes.data_as(C.c_void_p), cu.size, num_partitions))


class PackedBatch:
    """A packed ragged CSR/CSC batch resident in HBM. Keeps the torch tensors that back it alive."""

    def __init__(self, c_struct, tensors, raw_dev, has_intercept, join=None):
        self.c = c_struct
        self._join = join       # REDeviceSolver.pack_join of the context that packed it (gdmix_re_set_defer_unique), or None
        self._tensors = tensors
        self._raw_dev = raw_dev
        self.has_intercept = bool(has_intercept)
        self.E, self.N, self.Z, self.D = int(c_struct.E), int(c_struct.N), int(c_struct.Z), int(c_struct.D)
        self.P = self.D + (self.E if has_intercept else 0)
        self.max_p, self.max_n, self.max_nnz = int(c_struct.max_p), int(c_struct.max_n), int(c_struct.max_nnz)

    def _view(self, ptr, count, dtype):
        import torch
        ws = self._tensors["workspace"]
        off = ptr - ws.data_ptr()
        nbytes = count * torch.empty(0, dtype=dtype).element_size()
        return ws[off:off + nbytes].view(dtype)

    # device views (torch tensors) of the arrays produced by the pack kernels
    def ent_feat_ptr(self):
        import torch
        return self._view(self.c.ent_feat_ptr, self.E + 1, torch.int64)

    def ent_nnz_ptr(self):
        import torch
        return self._view(self.c.ent_nnz_ptr, self.E + 1, torch.int64)

    def unique_global(self):
        import torch
        if self._join is not None:
            self._join()        # the compaction that writes it may still be running next to the solve (a no-op once waited for)
        return self._view(self.c.unique_global, self.D, torch.int32)      # int32 since ABI 12: half the bytes to compact, to copy and to keep

    def __del__(self):  # pragma: no cover
        # the workspace goes back to torch's allocator in the order of the current stream: that stream must be behind a deferred
        # compaction nobody has waited for (a batch packed and dropped without a solve)
        try:
            if self._join is not None:
                self._join(dropping=True)
        except Exception:
            pass

    def csr_col(self):
        import torch
        return self._view(self.c.csr_col, self.Z, torch.int32)

    def row_ptr(self):
        import torch
        return self._view(self.c.row_ptr, self.N + self.E, torch.int32)

    def col_ptr(self):
        import torch
        return self._view(self.c.col_ptr, self.Z + self.E, torch.int32)   # d_e + 1 entries at ent_nnz_ptr[e] + e

    def csc_row(self):
        import torch
        return self._view(self.c.csc_row, self.Z, torch.int32)

    def csc_val(self):
        import torch
        return self._view(self.c.csc_val, self.Z, torch.float32)

    def coef_ptr_host(self):
        """[E+1] int64 numpy: offsets of each entity's coefficient slice in theta."""
        fp = self.ent_feat_ptr().cpu().numpy()
        return fp + (np.arange(self.E + 1, dtype=np.int64) if self.has_intercept else 0)


def host_array(t):
    """Device tensor -> numpy array. Large tensors are copied through page-locked memory (torch caches the blocks; the array
    keeps its block alive): about twice the rate of a copy into pageable memory."""
    if not getattr(t, "is_cuda", False):
        return t.cpu().numpy()
    if t.numel() * t.element_size() < (1 << 20):
        return t.cpu().numpy()
    import torch
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return h.numpy()


class SolveResult:
    def __init__(self, tensors, E, P):
        self._t = tensors
        self.E, self.P = E, P

    def __getattr__(self, k):
        t = self.__dict__.get("_t", {})
        if k in t:
            return t[k]
        raise AttributeError(k)

    def to_host(self, keys=None):
        """numpy copies of the result arrays (all of them, or only `keys`: the unthresholded coefficients are as large as
        the thresholded ones and the model path never reads them)."""
        out, staged = {}, []
        for k, v in self._t.items():
            if keys is not None and k not in keys:
                continue
            if v is None:
                out[k] = None
            elif v.is_cuda and v.numel() * v.element_size() >= (1 << 20):
                # large arrays go through page-locked memory (torch caches the blocks): about twice the rate of a copy
                # into pageable memory, and the copies of one result overlap
                import torch
                h = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                h.copy_(v, non_blocking=True)
                staged.append((k, h))
            else:
                out[k] = v.cpu().numpy()
        if staged:
            import torch
            torch.cuda.current_stream().synchronize()
            for k, h in staged:
                out[k] = h.numpy()    # the array keeps the pinned block alive
        return out


def _serialised(fn):
    """Calls on one context must be serialised by the caller (include/gdmix_re.h). The owner thread does that by construction; the one
    caller that is not the owner is a PackedBatch's destructor, which the garbage collector may run on any thread while the owner is
    inside gdmix_re_pack with the interpreter lock released: every context call of the solver takes its re-entrant lock."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with self._ctx_lock:
            return fn(self, *a, **k)
    return wrapper


class REDeviceSolver:
    """One MI355X: pack, solve and score entity batches through the C ABI."""

    def __init__(self, device: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise GdmixReError("no HIP device visible: the random-effect solve runs on MI355X only "
                               "(there is no CPU fallback)")
        self.torch = torch
        self.lib = load_library()
        self.tall_team_n = self.TALL_TEAM_N_DEFAULT     # (what gdmix_re_create sets; GDMIX_RE_TALL_TEAM=0 in the environment switches the class off)
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        torch.cuda.set_device(self.device)
        torch.zeros(1, device=self.device)  # force HIP context creation through torch's runtime
        h = C.c_void_p()
        import threading
        self._ctx_lock = threading.RLock()
        _check(self.lib.gdmix_re_create(self.device_index, C.byref(h)), "gdmix_re_create")
        self._h = h
        self._scratch = None
        # the compaction of the unique feature ids next to the solve (gdmix_re_set_defer_unique; GDMIX_RE_DEFER_UNIQUE=0: inside the pack)
        self._pack_gen = 0      # packs of this context so far: only the latest one's compaction can be outstanding
        self.set_defer_unique(os.environ.get("GDMIX_RE_DEFER_UNIQUE", "1") != "0")

    @_serialised
    def set_defer_unique(self, enabled: bool):
        _check(self.lib.gdmix_re_set_defer_unique(self._h, int(bool(enabled))), "gdmix_re_set_defer_unique")

    @_serialised
    def pack_join(self, stream=None):
        """A stream (default: the current one) waits for a pack's deferred compaction (PackedBatch.unique_global calls it; every
        library call that reads the array does so itself)."""
        if getattr(self, "_h", None):
            _check(self.lib.gdmix_re_pack_join(self._h, self._stream() if stream is None else stream), "gdmix_re_pack_join")

    @_serialised
    def _pack_join_of(self, gen, pack_stream, dropping=False):
        # a batch older than the context's latest pack has been waited for already (gdmix_re_pack waits first): its accessor or its
        # destructor must not make the stream wait for the NEWER batch's compaction, which is meant to run next to that batch's solve.
        # A reader waits on its own (the current) stream; a batch that is dropped makes the stream it was packed on wait — the one its
        # workspace goes back to the allocator on — whatever thread and stream the destructor happens to run under.
        if gen == self._pack_gen:
            self.pack_join(pack_stream if dropping else None)

    @_serialised
    def close(self):
        if getattr(self, "_h", None):
            self.lib.gdmix_re_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_wave_lds_limit(self, nbytes: int):
        """Entities whose LDS footprint exceeds nbytes go to the workgroup-per-entity kernel (0 = all)."""
        _check(self.lib.gdmix_re_set_wave_lds_limit(self._h, int(nbytes)), "set_wave_lds_limit")

    def set_kernel_mask(self, mask: int):
        """bit1: LDS-resident wave kernel, bit2: group kernels (the tall kernel and the team kernels are always on; bit0, the
        register wave kernel removed in round 4, is ignored). Default 7."""
        _check(self.lib.gdmix_re_set_kernel_mask(self._h, int(mask)), "set_kernel_mask")

    def set_giant_nnz(self, nnz: int):
        """Entities with >= nnz non-zeros are solved by the device-wide kernel (0 = never)."""
        _check(self.lib.gdmix_re_set_giant_nnz(self._h, int(nnz)), "set_giant_nnz")

    def set_team_nnz(self, nnz: int):
        """Entities with >= nnz non-zeros (below the giant threshold) are solved by the team tiers of the persistent kernel (0 = never)."""
        _check(self.lib.gdmix_re_set_team_nnz(self._h, int(nnz)), "set_team_nnz")

    TALL_MIN_N_DEFAULT = 32

    def set_tall_min_n(self, min_n: int):
        """Entities with at most 64 coefficients and at least min_n samples are solved by the tall kernel (0 = never)."""
        _check(self.lib.gdmix_re_set_tall_min_n(self._h, int(min_n)), "set_tall_min_n")

    def set_tall_split_n(self, split_n: int):
        """Tall entities of at least split_n samples get a CU each; smaller ones share a CU (eight single-wavefront workgroups)."""
        _check(self.lib.gdmix_re_set_tall_split_n(self._h, int(split_n)), "set_tall_split_n")

    TALL_TEAM_N_DEFAULT = 8192
    TALL_SPLIT_N_DEFAULT = 4096

    def pin_routing(self):
        """Make an entity's kernel a function of the entity alone. By default two thresholds are chosen per BATCH (where the
        one-CU-per-entity tall variant starts: 4096 / 2048 / 1024 / 512 samples; where the four-workgroup teams start: team_n x
        {1, 2, 4}), and those variants add an entity's sums in different orders — the same entity can come out with other last
        bits in another batch. Pinned: the split at its default whatever the batch holds, a team for every entity of at least
        TALL_TEAM_N_DEFAULT samples. For runs whose output must not depend on how entities are grouped into batches
        (entity re-balancing, comparisons across partitionings); it costs the per-batch tuning of latency-bound shares."""
        self.set_tall_split_n(self.TALL_SPLIT_N_DEFAULT)
        self.set_tall_team_n(-self.TALL_TEAM_N_DEFAULT)

    def set_tall_team_n(self, team_n: int):
        """The tallest entities of a batch get a team of four workgroups (four CUs of one XCD share the pass over one entity's samples).
        team_n > 0: eight-wavefront tall entities from team_n, 2 team_n or 4 team_n samples on — the lowest that keeps the class
        within one round of teams on the device; team_n < 0: every one of at least -team_n samples (tests); 0: never."""
        _check(self.lib.gdmix_re_set_tall_team_n(self._h, int(team_n)), "set_tall_team_n")
        self.tall_team_n = int(team_n)

    def device_shared(self) -> bool:
        """Is another PROCESS present on this solver's device right now (gdmix_re_device_shared)?"""
        rc = self.lib.gdmix_re_device_shared(self._h)
        if rc < 0:
            _check(rc, "gdmix_re_device_shared")
        return bool(rc)

    def set_tall_mid_n(self, mid_n: int):
        """The mid tall class (four wavefronts per entity, two workgroups per CU): mid_n < 0 per batch, > 0 every one-wavefront
        tall entity of at least mid_n samples (tests), 0 never (the default: profiles/r06_ml20m_mid.txt). gdmix_re_set_tall_mid_n."""
        _check(self.lib.gdmix_re_set_tall_mid_n(self._h, int(mid_n)), "set_tall_mid_n")

    def set_spread(self, queues: int):
        """Large size classes are dealt over `queues` streams (the caller's + the context's side streams; default 4) so that their tails
        overlap; 0: one after another on the caller's stream (per-class durations then do not stretch each other)."""
        _check(self.lib.gdmix_re_set_spread(self._h, int(queues)), "set_spread")

    def set_timing(self, enabled: bool):
        _check(self.lib.gdmix_re_set_timing(self._h, int(bool(enabled))), "set_timing")

    def last_solve_ms(self):
        """Per-size-class kernel milliseconds of the last solve (HIP events on the launch stream)."""
        ms = (C.c_float * NUM_CLASSES)()
        _check(self.lib.gdmix_re_last_solve_ms(self._h, ms), "last_solve_ms")
        return [float(x) for x in ms]

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    # ---- upload + pack -------------------------------------------------------------------------
    def upload(self, raw: RawBatch):
        """Copy the raw ragged arrays to HBM (pinned-free simple path). Returns dict of device tensors."""
        t = self.torch
        dev = self.device
        d = dict(ent_row_ptr=t.from_numpy(raw.ent_row_ptr).to(dev), row_nnz_ptr=t.from_numpy(raw.row_nnz_ptr).to(dev),
                 col_global=t.from_numpy(raw.col_global).to(dev), val=t.from_numpy(raw.val).to(dev),
                 y=t.from_numpy(raw.y).to(dev), offset=t.from_numpy(raw.offset).to(dev),
                 weight=None if raw.weight is None else t.from_numpy(raw.weight).to(dev))
        d["E"], d["N"], d["Z"] = raw.E, raw.N, raw.Z
        return d

    WIRE_ARRAYS = ("ent_n", "row_nnz", "col_global", "val", "y", "offset", "weight")

    def upload_wire(self, wire, pinned=None):
        """wire: RawBatch.to_wire() (host numpy). Copies the arrays to HBM on the current stream — through the page-locked
        tensors of `pinned` (same keys, at least as large; reused from partition to partition) when given, so that the
        copies are asynchronous. Returns the dict of device tensors gdmix_re_widen takes."""
        t = self.torch
        d = {k: wire[k] for k in ("E", "N", "Z", "row_nnz_width", "y_width", "col_width")}
        for k in self.WIRE_ARRAYS:
            a = wire[k]
            if a is None:
                d[k] = None
                continue
            if isinstance(a, t.Tensor):      # already a (page-locked) host tensor: the reader decoded into it
                d[k] = a.to(self.device, non_blocking=True)
                continue
            if pinned is not None:
                h = pinned[k][:a.size]
                h.numpy()[...] = a
                d[k] = h.to(self.device, non_blocking=True)
            else:
                d[k] = t.from_numpy(a).to(self.device)
        return d

    def wire_stage(self, wire):
        """Page-locked host tensors for upload_wire(wire, pinned=...), one per wire array, kept by the solver and grown when a
        partition needs more: the copies of the next partition reuse them (the previous upload must have completed: callers
        synchronise the stream once per partition anyway, when they read the results back)."""
        t = self.torch
        st = self.__dict__.setdefault("_wire_stage", {})
        for k in self.WIRE_ARRAYS:
            a = wire[k]
            if a is None or isinstance(a, t.Tensor):
                continue
            tdt = t.from_numpy(a[:0]).dtype
            cur = st.get(k)
            if cur is None or cur.dtype != tdt or cur.numel() < a.size:
                st[k] = t.empty(int(a.size * 1.25) + 1024, dtype=tdt, pin_memory=True)
        return st

    @_serialised
    def widen(self, wd):
        """gdmix_re_widen: wire form (device tensors from upload_wire) -> the raw form gdmix_re_pack takes, on the device."""
        t = self.torch
        E, N, Z = wd["E"], wd["N"], wd["Z"]
        ptr = lambda x: None if x is None or x.numel() == 0 else x.data_ptr()
        c_wire = _WireBatch(E, N, Z, ptr(wd["ent_n"]), ptr(wd["row_nnz"]), wd["row_nnz_width"], wd["y_width"], wd["col_width"], 0, ptr(wd["col_global"]),
                            ptr(wd["val"]), ptr(wd["y"]), ptr(wd["offset"]), ptr(wd["weight"]))
        nbytes = self.lib.gdmix_re_widen_workspace_bytes(E, N, Z)
        ws = t.empty(nbytes, dtype=t.uint8, device=self.device)
        c_raw = _RawBatch()
        _check(self.lib.gdmix_re_widen(self._h, C.byref(c_wire), ws.data_ptr(), nbytes, C.byref(c_raw), self._stream()), "gdmix_re_widen")

        def view(p, count, dtype):
            if not p or count == 0:
                return t.empty(0, dtype=dtype, device=self.device)
            off = p - ws.data_ptr()
            return ws[off:off + count * t.empty(0, dtype=dtype).element_size()].view(dtype)
        y = wd["y"] if wd["y_width"] == 4 else view(c_raw.y, N, t.float32)
        rd = dict(E=E, N=N, Z=Z, ent_row_ptr=view(c_raw.ent_row_ptr, E + 1, t.int64), row_nnz_ptr=view(c_raw.row_nnz_ptr, N + 1, t.int64),
                  col_global=view(c_raw.col_global, Z, t.int64), val=wd["val"], y=y, offset=wd["offset"], weight=wd["weight"],
                  _keep=(ws, wd))
        return rd

    @_serialised
    def pack(self, raw, has_intercept=True) -> PackedBatch:
        """raw: RawBatch (host) or the dict returned by upload() / widen() (device). A batch the reader narrowed (WireRawBatch) is uploaded
        in that form and widened on the device."""
        t = self.torch
        if isinstance(raw, WireRawBatch):
            rd = self.widen(self.upload_wire(raw.to_wire()))
        else:
            rd = self.upload(raw) if isinstance(raw, RawBatch) else raw
        E, N, Z = rd["E"], rd["N"], rd["Z"]
        ptr = lambda x: None if x is None or x.numel() == 0 else x.data_ptr()
        c_raw = _RawBatch(E, N, Z, rd["ent_row_ptr"].data_ptr(), rd["row_nnz_ptr"].data_ptr(), ptr(rd["col_global"]),
                          ptr(rd["val"]), ptr(rd["y"]), ptr(rd["offset"]), ptr(rd["weight"]))
        nbytes = self.lib.gdmix_re_pack_workspace_bytes(E, N, Z)
        ws = t.empty(nbytes, dtype=t.uint8, device=self.device)
        c_packed = _Packed()
        st = self._stream()
        _check(self.lib.gdmix_re_pack(self._h, C.byref(c_raw), int(bool(has_intercept)), ws.data_ptr(), nbytes,
                                      C.byref(c_packed), st), "gdmix_re_pack")
        self._pack_gen += 1
        return PackedBatch(c_packed, {"workspace": ws}, rd, has_intercept, join=functools.partial(self._pack_join_of, self._pack_gen, st))

    # ---- solve -----------------------------------------------------------------------------------
    def alloc_result(self, packed: PackedBatch, variance=False):
        t, dev = self.torch, self.device
        E, P = packed.E, packed.P
        return dict(theta=t.empty(P, dtype=t.float64, device=dev), theta_thr=t.empty(P, dtype=t.float64, device=dev),
                    variance=t.empty(P, dtype=t.float64, device=dev) if variance else None,
                    fval=t.empty(E, dtype=t.float64, device=dev), gnorm=t.empty(E, dtype=t.float64, device=dev),
                    nit=t.empty(E, dtype=t.int32, device=dev), nfev=t.empty(E, dtype=t.int32, device=dev),
                    status=t.full((E,), -1, dtype=t.int32, device=dev))

    @_serialised
    def solve(self, packed: PackedBatch, opts: Optional[SolverOptions] = None, theta0=None, out=None) -> SolveResult:
        t = self.torch
        opts = opts or SolverOptions()
        if bool(opts.has_intercept) != packed.has_intercept:
            raise GdmixReError("has_intercept of the options differs from the packed batch")
        c_opts = opts.to_c()
        need = self.lib.gdmix_re_solve_scratch_bytes(C.byref(packed.c), C.byref(c_opts))
        if need > packed.c.scratch_bytes and (self._scratch is None or self._scratch.numel() < need):
            self._scratch = t.empty(need, dtype=t.uint8, device=self.device)
        if self._scratch is not None:
            _check(self.lib.gdmix_re_set_scratch(self._h, self._scratch.data_ptr(), self._scratch.numel()), "set_scratch")
        if theta0 is not None:
            if isinstance(theta0, np.ndarray):
                theta0 = t.from_numpy(np.ascontiguousarray(theta0, np.float64)).to(self.device)
            elif not theta0.is_cuda:   # a host tensor, page-locked if the caller wants the copy to be asynchronous
                theta0 = theta0.to(self.device, non_blocking=theta0.is_pinned())
            if theta0.numel() != packed.P or theta0.dtype != t.float64:
                raise GdmixReError("theta0 must be float64 with one entry per coefficient")
        tensors = out or self.alloc_result(packed, variance=c_opts.variance_mode != VAR_NONE)
        p = lambda x: None if x is None else x.data_ptr()
        c_res = _Result(p(tensors["theta"]), p(tensors["theta_thr"]), p(tensors.get("variance")), p(tensors["fval"]),
                        p(tensors["gnorm"]), p(tensors["nit"]), p(tensors["nfev"]), p(tensors["status"]))
        _check(self.lib.gdmix_re_solve(self._h, C.byref(packed.c), C.byref(c_opts), p(theta0), C.byref(c_res),
                                       self._stream()), "gdmix_re_solve")
        return SolveResult(tensors, packed.E, packed.P)

    @_serialised
    def variance_full(self, packed: PackedBatch, opts: SolverOptions, theta):
        """diag of the inverse Hessian of every entity at theta ([P], local index space; numpy or device tensor) -> device tensor [P]."""
        t = self.torch
        if isinstance(theta, np.ndarray):
            theta = t.from_numpy(np.ascontiguousarray(theta, np.float64)).to(self.device)
        if theta.numel() != packed.P or theta.dtype != t.float64:
            raise GdmixReError("theta must be float64 with one entry per coefficient")
        c_opts = opts.to_c()
        c_opts.variance_mode = VAR_FULL
        need = self.lib.gdmix_re_solve_scratch_bytes(C.byref(packed.c), C.byref(c_opts))
        if need > packed.c.scratch_bytes and (self._scratch is None or self._scratch.numel() < need):
            self._scratch = t.empty(need, dtype=t.uint8, device=self.device)
        if self._scratch is not None:
            _check(self.lib.gdmix_re_set_scratch(self._h, self._scratch.data_ptr(), self._scratch.numel()), "set_scratch")
        out = t.empty(packed.P, dtype=t.float64, device=self.device)
        _check(self.lib.gdmix_re_variance_full(self._h, C.byref(packed.c), C.byref(c_opts), theta.data_ptr(), out.data_ptr(), self._stream()),
               "gdmix_re_variance_full")
        return out

    # ---- fixed effect, FULL variances with several workers (include/gdmix_fe.h) -----------------------
    @_serialised
    def hessian_dense(self, packed: PackedBatch, theta_local, has_intercept=True):
        """X~' D X~ of a one-entity batch (a worker's shard) at theta_local (the shard's local order, intercept first) as a dense
        [ld, ld] device tensor, ld = p rounded up to 64; no regulariser."""
        t = self.torch
        if isinstance(theta_local, np.ndarray):
            theta_local = t.from_numpy(np.ascontiguousarray(theta_local, np.float64)).to(self.device)
        p = packed.D + (1 if has_intercept else 0)
        ld = (p + 63) // 64 * 64
        H = t.empty((ld, ld), dtype=t.float64, device=self.device)
        nbytes = self.lib.gdmix_fe_hessian_dense_scratch_bytes(C.byref(packed.c))
        scratch = t.empty(nbytes, dtype=t.uint8, device=self.device)
        _check(self.lib.gdmix_fe_hessian_dense(self._h, C.byref(packed.c), int(bool(has_intercept)), theta_local.data_ptr(), H.data_ptr(), ld,
                                               scratch.data_ptr(), nbytes, self._stream()), "gdmix_fe_hessian_dense")
        return H

    @_serialised
    def variance_of_hessian(self, H, p, l2, unregularised_index=-1):
        """diag((H + (l2 + 1e-12) I - l2 e_u e_u')^-1)[0, p) of a summed curvature matrix H ([ld, ld] device tensor, overwritten)."""
        t = self.torch
        ld = int(H.shape[0])
        work = t.empty_like(H)
        out = t.empty(int(p), dtype=t.float64, device=self.device)
        _check(self.lib.gdmix_fe_variance_of_hessian(self._h, H.data_ptr(), int(p), ld, float(l2), int(unregularised_index), work.data_ptr(),
                                                     out.data_ptr(), self._stream()), "gdmix_fe_variance_of_hessian")
        return out

    def class_counts(self, packed: PackedBatch):
        """Entities per size class of the last solve on this batch (host list) + kernel names."""
        t = self.torch
        cc = packed._view(packed.c.class_count, NUM_CLASSES, t.int32).cpu().numpy()
        names = [self.lib.gdmix_re_class_kernel_name(c).decode() for c in range(NUM_CLASSES)]
        return list(zip(names, cc[:NUM_CLASSES].tolist()))

    # ---- score -----------------------------------------------------------------------------------
    @_serialised
    def score(self, packed: PackedBatch, theta, has_model=None):
        t = self.torch
        if isinstance(theta, np.ndarray):
            theta = t.from_numpy(np.ascontiguousarray(theta, np.float64)).to(self.device)
        if has_model is not None and isinstance(has_model, np.ndarray):
            has_model = t.from_numpy(np.ascontiguousarray(has_model, np.uint8)).to(self.device)
        logit = t.empty(packed.N, dtype=t.float32, device=self.device)
        per = t.empty(packed.N, dtype=t.float32, device=self.device)
        _check(self.lib.gdmix_re_score(self._h, C.byref(packed.c), int(packed.has_intercept), theta.data_ptr(),
                                       None if has_model is None else has_model.data_ptr(), logit.data_ptr(),
                                       per.data_ptr(), self._stream()), "gdmix_re_score")
        return logit, per

    def partition_ids(self, ids, num_partitions: int):
        """Math.abs(id.toString.hashCode) % numPartitions for int64 entity ids, on the device."""
        t = self.torch
        if isinstance(ids, np.ndarray):
            ids = t.from_numpy(np.ascontiguousarray(ids, np.int64)).to(self.device)
        out = t.empty(ids.numel(), dtype=t.int32, device=self.device)
        _check(self.lib.gdmix_java_partition_ids_i64(self._h, ids.data_ptr(), ids.numel(), int(num_partitions),
                                                     out.data_ptr(), self._stream()), "partition_ids")
        return out
