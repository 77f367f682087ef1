"""ctypes front-end of oracle/libre_oracle.so — the CPU restatement of the reference RE hot path.

TEST INFRASTRUCTURE ONLY (see oracle/re_oracle.c header): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg — never by gdmix_amd/.
Pinned twice: against the fixtures the reference itself produced (tests/test_oracle_golden.py, tests/golden/) and against
scipy.optimize.fmin_l_bfgs_b run live on seeded problems (tests/test_oracle_scipy.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libre_oracle.so")
_lib = None


class OracleOpts(C.Structure):
    # same field order as gdmix_re_opts (include/gdmix_re.h)
    _fields_ = [("l2", C.c_double), ("regularize_bias", C.c_int32), ("has_intercept", C.c_int32),
                ("m", C.c_int32), ("max_iter", C.c_int32), ("maxfun", C.c_int32), ("maxls", C.c_int32),
                ("ftol", C.c_double), ("pgtol", C.c_double), ("variance_mode", C.c_int32),
                ("threshold", C.c_double), ("sum_loss", C.c_int32), ("linear", C.c_int32)]


def make_opts(l2=1.0, regularize_bias=True, has_intercept=True, m=10, max_iter=100, maxfun=15000,
              maxls=20, ftol=1e-12, pgtol=1e-5, variance_mode=0, threshold=1e-4, sum_loss=False, linear=False):
    return OracleOpts(float(l2), int(bool(regularize_bias)), int(bool(has_intercept)), int(m),
                      int(max_iter), int(maxfun), int(maxls), float(ftol), float(pgtol),
                      int(variance_mode), float(threshold), int(bool(sum_loss)), int(bool(linear)))


def build(force=False):
    """Compile oracle/libre_oracle.so with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "re_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_pack.restype = C.c_int64
        _lib.oracle_solve.restype = C.c_int
        _lib.oracle_solve2.restype = C.c_int
        _lib.oracle_score.restype = C.c_int
        _lib.oracle_java_string_hash.restype = C.c_int32
        _lib.oracle_java_partition_id.restype = C.c_int32
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def pack(ent_row_ptr, row_nnz_ptr, col_global):
    """np.unique per entity -> dict(ent_nnz_ptr, ent_feat_ptr, row_ptr, csr_col, unique_global, D)."""
    ent_row_ptr = _c(ent_row_ptr, np.int64)
    row_nnz_ptr = _c(row_nnz_ptr, np.int64)
    col_global = _c(col_global, np.int64)
    E = ent_row_ptr.size - 1
    N = row_nnz_ptr.size - 1
    Z = col_global.size
    ent_nnz_ptr = np.zeros(E + 1, np.int64)
    ent_feat_ptr = np.zeros(E + 1, np.int64)
    row_ptr = np.zeros(N + E, np.int32)
    csr_col = np.zeros(max(Z, 1), np.int32)
    unique_global = np.zeros(max(Z, 1), np.int64)
    D = lib().oracle_pack(C.c_int64(E), _p(ent_row_ptr), _p(row_nnz_ptr), _p(col_global),
                          _p(ent_nnz_ptr), _p(ent_feat_ptr), _p(row_ptr), _p(csr_col), _p(unique_global))
    if D < 0:
        raise MemoryError("oracle_pack failed")
    return dict(E=E, N=N, Z=Z, D=int(D), ent_row_ptr=ent_row_ptr, ent_nnz_ptr=ent_nnz_ptr,
                ent_feat_ptr=ent_feat_ptr, row_ptr=row_ptr, csr_col=csr_col[:Z],
                unique_global=unique_global[:int(D)])


def solve(packed, val, y, offset, weight=None, opts=None, theta0=None, e_begin=0, e_end=None):
    """Run the fp64 L-BFGS restatement on entities [e_begin, e_end) of a packed batch. `nfev` is scipy's funcalls (a trial
    point equal to the previously evaluated one is served from ScalarFunction's cache and not counted); `neval` counts every
    evaluation performed, which is what the device reports — the two differ only when a step is too small to move x."""
    opts = opts or make_opts()
    E = packed["E"]
    e_end = E if e_end is None else e_end
    ic = 1 if opts.has_intercept else 0
    P = packed["D"] + E * ic
    val = _c(val, np.float32)
    y = _c(y, np.float32)
    offset = _c(offset, np.float32)
    weight = _c(weight, np.float32)
    theta0 = _c(theta0, np.float64)
    out = dict(theta=np.zeros(P), theta_thr=np.zeros(P),
               variance=np.zeros(P) if opts.variance_mode else None,
               fval=np.zeros(E), gnorm=np.zeros(E), nit=np.zeros(E, np.int32),
               nfev=np.zeros(E, np.int32), status=np.full(E, -1, np.int32), neval=np.zeros(E, np.int32))
    rc = lib().oracle_solve2(C.c_int64(e_begin), C.c_int64(e_end), _p(packed["ent_row_ptr"]),
                            _p(packed["ent_nnz_ptr"]), _p(packed["ent_feat_ptr"]), _p(packed["row_ptr"]),
                            _p(packed["csr_col"]), _p(val), _p(y), _p(offset), _p(weight),
                            C.byref(opts), _p(theta0), _p(out["theta"]), _p(out["theta_thr"]),
                            _p(out["variance"]), _p(out["fval"]), _p(out["gnorm"]), _p(out["nit"]),
                            _p(out["nfev"]), _p(out["status"]), _p(out["neval"]))
    if rc:
        raise RuntimeError(f"oracle_solve failed: {rc}")
    return out


def branch_counts(reset=True):
    """Which L-BFGS-B branches the solves of this thread went through since the last reset:
    dict(skipped_pairs, gd_restarts, maxls_aborts, max_evals_in_one_search, memory_wraps)."""
    out = (C.c_longlong * 5)()
    lib().oracle_branch_counts(out, int(bool(reset)))
    return dict(zip(("skipped_pairs", "gd_restarts", "maxls_aborts", "max_evals_in_one_search", "memory_wraps"), [int(v) for v in out]))


def score(packed, val, offset, theta, has_intercept=True, has_model=None):
    val = _c(val, np.float32)
    offset = _c(offset, np.float32)
    theta = _c(theta, np.float64)
    has_model = _c(has_model, np.uint8)
    N = packed["N"]
    logit = np.zeros(N, np.float32)
    per_coord = np.zeros(N, np.float32)
    rc = lib().oracle_score(C.c_int64(packed["E"]), _p(packed["ent_row_ptr"]), _p(packed["ent_nnz_ptr"]),
                            _p(packed["ent_feat_ptr"]), _p(packed["row_ptr"]), _p(packed["csr_col"]),
                            _p(val), _p(offset), C.c_int(int(bool(has_intercept))), _p(theta),
                            _p(has_model), _p(logit), _p(per_coord))
    if rc:
        raise RuntimeError(f"oracle_score failed: {rc}")
    return logit, per_coord


def _utf16(s):
    return np.frombuffer(s.encode("utf-16-le"), dtype=np.uint16).copy()


def java_string_hash(s):
    cu = _utf16(s)
    return int(lib().oracle_java_string_hash(_p(cu), C.c_int64(cu.size)))


def java_partition_id(s, num_partitions):
    cu = _utf16(s)
    return int(lib().oracle_java_partition_id(_p(cu), C.c_int64(cu.size), C.c_int32(num_partitions)))
