"""ctypes binding of libgdmix_io.so (include/gdmix_io.h): the native, multi-threaded reader of entity-grouped
TFRecord partitions. Same rules and same errors as grouped_reader.read_grouped_partition (which stays as the
statement of those rules in Python); tests/test_native_io.py compares the two array for array."""
import ctypes as C
from collections.abc import Sequence
import os
import threading

import numpy as np

from ..batch import RawBatch, WireRawBatch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_HERE, "libgdmix_io.so")
ABI_VERSION = 7
EXPORTED_SYMBOLS = ("gdmix_io_abi_version", "gdmix_io_build_id", "gdmix_io_last_error", "gdmix_io_read_grouped", "gdmix_io_free", "gdmix_io_narrow", "gdmix_io_pool_trim",
                    "gdmix_io_crc32c", "gdmix_io_masked_crc32c", "gdmix_io_avro_write_models", "gdmix_io_avro_write_scores",
                    "gdmix_io_write_grouped", "gdmix_io_read_examples", "gdmix_io_avro_read_models", "gdmix_io_free_models", "gdmix_io_map_coefficients", "gdmix_io_ids_unique", "gdmix_io_match_ids")


class GdmixIoError(RuntimeError):
    pass


class _Schema(C.Structure):
    _fields_ = [("entity", C.c_char_p), ("feature_bag", C.c_char_p), ("offset", C.c_char_p), ("uid", C.c_char_p),
                ("label", C.c_char_p), ("weight", C.c_char_p), ("num_features", C.c_int64), ("check_crc", C.c_int32),
                ("threads", C.c_int32)]


class _Batch(C.Structure):
    _fields_ = [("E", C.c_int64), ("N", C.c_int64), ("Z", C.c_int64), ("ent_row_ptr", C.POINTER(C.c_int64)),
                ("row_nnz_ptr", C.POINTER(C.c_int64)), ("col_global", C.POINTER(C.c_int64)), ("val", C.POINTER(C.c_float)),
                ("y", C.POINTER(C.c_float)), ("offset", C.POINTER(C.c_float)), ("weight", C.POINTER(C.c_float)),
                ("uid", C.POINTER(C.c_int64)), ("ent_id_ptr", C.POINTER(C.c_int64)), ("ent_id_bytes", C.POINTER(C.c_char)),
                ("has_label", C.c_int32), ("bytes_read", C.c_int64), ("labels_binary", C.c_int32),
                ("ent_n", C.POINTER(C.c_int32)), ("row_nnz", C.c_void_p), ("col", C.c_void_p), ("y8", C.POINTER(C.c_uint8)),
                ("row_nnz_width", C.c_int32), ("col_width", C.c_int32)]


class _ModelTable(C.Structure):
    _fields_ = [("E", C.c_int64), ("id_ptr", C.c_void_p), ("id_bytes", C.c_char_p), ("coef_beg", C.c_void_p),
                ("coef_cnt", C.c_void_p), ("var_beg", C.c_void_p), ("feat_beg", C.c_void_p), ("mean", C.c_void_p),
                ("variance", C.c_void_p), ("feat_idx", C.c_void_p), ("prefix_ptr", C.c_void_p), ("prefix_bytes", C.c_char_p),
                ("n_prefix", C.c_int64), ("icpt_enc", C.c_char_p), ("icpt_len", C.c_int64), ("class_enc", C.c_char_p),
                ("class_len", C.c_int64), ("loss_enc", C.c_char_p), ("loss_len", C.c_int64), ("has_intercept", C.c_int32),
                ("threshold", C.c_double)]


class _Models(C.Structure):
    _fields_ = [("E", C.c_int64), ("C", C.c_int64), ("id_ptr", C.POINTER(C.c_int64)), ("id_bytes", C.POINTER(C.c_char)),
                ("coef_ptr", C.POINTER(C.c_int64)), ("mean", C.POINTER(C.c_double)), ("variance", C.POINTER(C.c_double)),
                ("F", C.c_int64), ("feat_idx", C.POINTER(C.c_int64)), ("has_variance", C.POINTER(C.c_uint8)), ("any_variance", C.c_int32)]


_lib = None

# Threads per native call while a host pipeline is open (model.begin_pipeline): several decodes and writers run at once there, and
# the library's default — the optimum of ONE call alone — oversubscribes the box. A module variable with a lock, scoped by
# pipeline_threads_begin / _end: not the process environment (an embedding host's other calls keep their defaults, and nothing
# calls setenv next to native threads that are inside getenv).
_pipeline_threads = []
_pipeline_lock = threading.Lock()


def pipeline_threads_begin(count: int) -> None:
    with _pipeline_lock:
        _pipeline_threads.append(max(1, int(count)))


def pipeline_threads_end() -> None:
    with _pipeline_lock:
        if _pipeline_threads:
            _pipeline_threads.pop()


def _threads(threads) -> int:
    """An explicit count wins; then an open pipeline's; 0 = the library's own default (GDMIX_IO_THREADS or its built-in)."""
    t = int(threads)
    if t > 0 or not _pipeline_threads or os.environ.get("GDMIX_IO_THREADS"):
        return t
    return _pipeline_threads[-1]


def available() -> bool:
    return os.path.exists(LIB_PATH)


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GdmixIoError(f"{LIB_PATH} is missing: run `python -m gdmix_amd.build`")
    from .. import build as _build
    stale = _build.check_library(LIB_PATH, _build.io_source_id, "csrc/io_*.cpp")
    if stale:
        raise GdmixIoError(stale)
    lib = C.CDLL(LIB_PATH)
    for sym in EXPORTED_SYMBOLS:
        if not hasattr(lib, sym):
            raise GdmixIoError(f"{LIB_PATH} does not export {sym}")
    lib.gdmix_io_abi_version.restype = C.c_int
    lib.gdmix_io_last_error.restype = C.c_char_p
    lib.gdmix_io_build_id.restype = C.c_char_p
    lib.gdmix_io_read_grouped.argtypes = [C.POINTER(C.c_char_p), C.c_int32, C.POINTER(_Schema), C.POINTER(C.POINTER(_Batch))]
    lib.gdmix_io_read_examples.argtypes = [C.POINTER(C.c_char_p), C.c_int32, C.POINTER(_Schema), C.POINTER(C.POINTER(_Batch))]
    lib.gdmix_io_free.argtypes = [C.POINTER(_Batch)]
    lib.gdmix_io_free.restype = None
    lib.gdmix_io_narrow.argtypes = [C.POINTER(_Batch), C.c_int32]
    lib.gdmix_io_pool_trim.argtypes = []
    lib.gdmix_io_pool_trim.restype = C.c_size_t
    lib.gdmix_io_avro_write_models.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_char_p, C.POINTER(_ModelTable),
                                               C.c_int32, C.c_int32, C.c_int32]
    lib.gdmix_io_avro_write_scores.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.gdmix_io_avro_read_models.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int32, C.c_void_p, C.c_char_p, C.c_int64,
                                              C.c_char_p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.POINTER(_Models))]
    lib.gdmix_io_free_models.argtypes = [C.POINTER(_Models)]
    lib.gdmix_io_map_coefficients.argtypes = [C.c_int64] + [C.c_void_p] * 7 + [C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
    lib.gdmix_io_free_models.restype = None
    lib.gdmix_io_ids_unique.argtypes = [C.c_char_p, C.c_void_p, C.c_int64]
    lib.gdmix_io_match_ids.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
    lib.gdmix_io_write_grouped.argtypes = [C.c_char_p, C.POINTER(_Batch), C.POINTER(_Schema), C.c_int32]
    lib.gdmix_io_crc32c.argtypes = [C.c_char_p, C.c_size_t]
    lib.gdmix_io_crc32c.restype = C.c_uint32
    lib.gdmix_io_masked_crc32c.argtypes = [C.c_char_p, C.c_size_t]
    lib.gdmix_io_masked_crc32c.restype = C.c_uint32
    if lib.gdmix_io_abi_version() != ABI_VERSION:
        raise GdmixIoError(f"{LIB_PATH}: ABI version {lib.gdmix_io_abi_version()}, expected {ABI_VERSION}")
    _lib = lib
    return lib


def pool_trim() -> int:
    """Release the idle array blocks and writer buffers the library keeps between partitions (include/gdmix_io.h,
    gdmix_io_pool_trim). Returns the bytes released; 0 if the library is not loaded."""
    return int(_lib.gdmix_io_pool_trim()) if _lib is not None else 0


def crc32c(data: bytes) -> int:
    return int(load_library().gdmix_io_crc32c(data, len(data)))


def masked_crc32c(data: bytes) -> int:
    return int(load_library().gdmix_io_masked_crc32c(data, len(data)))


def _enc(s):
    return None if s is None else s.encode("utf-8")


def _copy(ptr, count, dtype):
    if count == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)


class _BatchOwner:
    """Keeps a gdmix_io_batch alive for as long as a numpy view of one of its arrays is; frees it afterwards."""

    def __init__(self, lib, handle, free=None):
        self._free, self._handle = free or lib.gdmix_io_free, handle

    def __del__(self):
        if self._handle is not None:
            self._free(self._handle)
            self._handle = None


_CTYPE = {np.dtype(np.int64): C.c_int64, np.dtype(np.float32): C.c_float, np.dtype(np.float64): C.c_double, np.dtype(np.int32): C.c_int32,
          np.dtype(np.uint8): C.c_uint8, np.dtype(np.uint16): C.c_uint16, np.dtype(np.uint32): C.c_uint32}


def _view(owner, ptr, count, dtype):
    """The library's array as a numpy array without a copy (hundreds of MB per partition); the array holds the owner."""
    if count == 0:
        return np.zeros(0, dtype)
    buf = (_CTYPE[np.dtype(dtype)] * count).from_address(ptr if isinstance(ptr, int) else C.addressof(ptr.contents))
    buf._owner = owner
    return np.frombuffer(buf, dtype=dtype)


class EntityIds(Sequence):
    """The entity ids of a decoded partition: concatenated UTF-8 bytes + offsets as the native reader returns them. It
    behaves like the list of str the Python reader returns, but only becomes one when somebody indexes or iterates it:
    a cold training run hands the bytes straight to the model-file writer and a million ids never become a million
    Python objects."""

    def __init__(self, raw: bytes, ptr: np.ndarray):
        self.raw = raw
        self.ptr = np.ascontiguousarray(ptr, np.int64)
        self._list = None
        self._unique = None

    def tolist(self):
        if self._list is None:
            self._list = _split_ids(self.raw, self.ptr, len(self))
        return self._list

    def __len__(self):
        return int(self.ptr.size) - 1

    def __getitem__(self, i):
        return self.tolist()[i]

    def __iter__(self):
        return iter(self.tolist())

    def __eq__(self, other):
        if isinstance(other, EntityIds):
            return self.raw == other.raw and np.array_equal(self.ptr, other.ptr)
        return self.tolist() == (other.tolist() if hasattr(other, "tolist") else list(other))

    def __repr__(self):
        return f"EntityIds({len(self)} ids)"

    def rows_in(self, table_ids: "EntityIds", threads=0) -> np.ndarray:
        """For every id of self its position in table_ids (whose ids are all different), or -1."""
        out = np.empty(len(self), np.int64)
        rc = load_library().gdmix_io_match_ids(table_ids.raw, table_ids.ptr.ctypes.data, len(table_ids), self.raw, self.ptr.ctypes.data,
                                               len(self), out.ctypes.data, _threads(threads))
        if rc != 0:
            raise GdmixIoError("gdmix_io_match_ids: " + load_library().gdmix_io_last_error().decode("utf-8", "replace"))
        return out

    def take(self, rows) -> "EntityIds":
        """The ids at `rows`, in that order."""
        from ..batch import _ranges
        rows = np.asarray(rows, np.int64)
        lens = self.ptr[rows + 1] - self.ptr[rows]
        ptr = np.zeros(rows.size + 1, np.int64)
        np.cumsum(lens, out=ptr[1:])
        raw = np.frombuffer(self.raw, np.uint8)[_ranges(self.ptr[rows], lens)].tobytes() if rows.size else b""
        out = EntityIds(raw, ptr)
        out._unique = True if self._unique else None
        return out

    def extended(self, other: "EntityIds") -> "EntityIds":
        """self followed by other."""
        return EntityIds(self.raw + other.raw, np.concatenate([self.ptr, self.ptr[-1] + other.ptr[1:]]))

    def all_different(self) -> bool:
        if self._unique is None:
            rc = load_library().gdmix_io_ids_unique(self.raw, self.ptr.ctypes.data, len(self))
            if rc < 0:
                raise GdmixIoError("gdmix_io_ids_unique failed")
            self._unique = bool(rc)
        return self._unique


def _split_ids(raw, ptr, E):
    """E strings from concatenated UTF-8 bytes + offsets, without a Python-level loop when possible."""
    if E == 0:
        return []
    lens = np.diff(ptr)
    if lens.min() > 0 and b"\x00" not in raw:
        # insert a NUL after every id, decode once, split once
        out = np.zeros(len(raw) + E, np.uint8)
        keep = np.ones(len(raw) + E, bool)
        keep[ptr[1:] + np.arange(E)] = False
        out[keep] = np.frombuffer(raw, np.uint8)
        return out.tobytes().decode("utf-8").split("\x00")[:E]
    return [raw[ptr[i]:ptr[i + 1]].decode("utf-8") for i in range(E)]


def read_grouped_files(files, entity_name, feature_bag, offset_column_name, uid_column_name, label_column_name=None,
                       weight_column_name=None, num_features=None, check_crc=False, threads=0, stats=None, wire=False) -> RawBatch:
    """files -> RawBatch; same arguments as grouped_reader.read_grouped_partition after file resolution and the
    metadata checks. weight_column_name None => no weight array. wire=True: the library narrows the partition to the 32-bit hand-over
    form (gdmix_io_narrow) and the result is a batch.WireRawBatch around its arrays — what a device solver uploads as it is."""
    lib = load_library()
    sc = _Schema(_enc(entity_name), _enc(feature_bag), _enc(offset_column_name), _enc(uid_column_name),
                 _enc(label_column_name), _enc(weight_column_name),
                 -1 if num_features is None or feature_bag is None else int(num_features), int(bool(check_crc)), _threads(threads))
    arr = (C.c_char_p * len(files))(*[f.encode("utf-8") for f in files])
    out = C.POINTER(_Batch)()
    rc = lib.gdmix_io_read_grouped(arr, len(files), C.byref(sc), C.byref(out))
    if rc != 0:
        msg = lib.gdmix_io_last_error().decode("utf-8", "replace")
        # same exception types as the Python reader: schema problems are KeyError / ValueError / AssertionError there
        raise (ValueError if rc in (-3, -4) else GdmixIoError)(f"gdmix_io_read_grouped: {msg}")
    owner = _BatchOwner(lib, out)     # freed when the last array view goes away
    b = out.contents
    E, N, Z = int(b.E), int(b.N), int(b.Z)
    idp = _copy(b.ent_id_ptr, E + 1, np.int64)
    raw_ids = C.string_at(b.ent_id_bytes, int(idp[-1])) if E else b""
    ids = EntityIds(raw_ids, idp)
    if stats is not None:
        stats["bytes_read"] = int(b.bytes_read)
    v = lambda ptr, n, dt: _view(owner, ptr, n, dt)
    if bool(b.has_label) and not int(b.labels_binary):
        raise AssertionError("labels must be 0 or 1")   # fit() asserts it (binary_logistic_regression.py:208)
    if wire:
        rc = lib.gdmix_io_narrow(out, _threads(threads))
        if rc != 0:
            raise (ValueError if rc == -6 else GdmixIoError)(f"gdmix_io_narrow: {lib.gdmix_io_last_error().decode('utf-8', 'replace')}")
        kdt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[int(b.row_nnz_width)]
        cdt = {2: np.uint16, 4: np.int32}[int(b.col_width)]
        return WireRawBatch(ent_row_ptr=v(b.ent_row_ptr, E + 1, np.int64), ent_n=v(b.ent_n, E, np.int32),
                            row_nnz=v(b.row_nnz or 0, N, kdt), col=v(b.col or 0, Z, cdt), val=v(b.val, Z, np.float32),
                            y=v(b.y, N, np.float32), y8=v(b.y8, N, np.uint8) if b.y8 else None, offset=v(b.offset, N, np.float32),
                            weight=v(b.weight, N, np.float32) if weight_column_name is not None else None,
                            uid=v(b.uid, N, np.int64), entity_ids=ids, has_label=bool(b.has_label))
    # the library built these arrays itself (monotone pointers, matching lengths, labels checked while decoding): the
    # passes RawBatch.validate would make over them are skipped
    return RawBatch(ent_row_ptr=v(b.ent_row_ptr, E + 1, np.int64), row_nnz_ptr=v(b.row_nnz_ptr, N + 1, np.int64),
                    col_global=v(b.col_global, Z, np.int64), val=v(b.val, Z, np.float32),
                    y=v(b.y, N, np.float32), offset=v(b.offset, N, np.float32),
                    weight=v(b.weight, N, np.float32) if weight_column_name is not None else None,
                    uid=v(b.uid, N, np.int64), entity_ids=ids, has_label=bool(b.has_label), trusted=True)


# ---- Avro writers ------------------------------------------------------------------------------------------
def _ptr(a):
    return None if a is None else a.ctypes.data


def _ids_to_bytes(ids):
    """Concatenated UTF-8 bytes of the ids + [E+1] offsets. Joined, encoded and measured in one piece when no id contains
    the separator (a million Python-level encode calls otherwise)."""
    if isinstance(ids, EntityIds):
        return ids.raw, ids.ptr
    E = len(ids)
    ptr = np.zeros(E + 1, np.int64)
    if E == 0:
        return b"", ptr
    try:
        joined = "\x00".join(ids)
    except TypeError:
        joined = "\x00".join(str(x) for x in ids)
    raw = joined.encode("utf-8")
    sep = np.flatnonzero(np.frombuffer(raw, np.uint8) == 0)
    if sep.size == E - 1:
        ptr[1:E] = sep - np.arange(E - 1)
        ptr[E] = len(raw) - (E - 1)
        return raw.replace(b"\x00", b""), ptr
    enc = [str(x).encode("utf-8") for x in ids]
    np.cumsum([len(x) for x in enc], out=ptr[1:])
    return b"".join(enc), ptr


class EncodedFeatures:
    """The feature list of a model file, encoded once: per global feature index the Avro bytes string(name) + string(term), as one
    byte string + offsets — what the model writer and reader hand to the library. Built per feature file, not per model file: with
    65 536 features (C5) the list comprehension, the cumsum and the join are ~0.1 s of interpreter time holding the lock, once per
    PARTITION before round 4 (profiles/r04_host_path.txt: the model Avro of a C5-shaped run 0.63 -> s of thread time)."""

    def __init__(self, prefix):
        prefix = list(prefix)
        self.count = len(prefix)
        self.ptr = np.zeros(self.count + 1, np.int64)
        if prefix:
            np.cumsum(np.fromiter((len(x) for x in prefix), np.int64, self.count), out=self.ptr[1:])
        self.bytes = b"".join(prefix)

    @classmethod
    def of(cls, prefix):
        return prefix if isinstance(prefix, cls) else cls(prefix)

    @classmethod
    def from_feature_file(cls, path):
        """The same straight from the bytes of a feature file (`name,term` per line, io/features.py), without a Python object per
        feature: for a plain file — no quotes, no carriage returns, one comma per line, names and terms shorter than 64 bytes — the
        encoding string(name) + string(term) = [2 len(name)] name [2 len(term)] term is exactly as long as the line `name,term\n`, so
        the encoded list is the file shifted right by one byte with the two length bytes dropped on the previous newline and on the
        comma. Returns None when the file is not that plain (the caller encodes it feature by feature). 65 536 features: 0.3 ms
        instead of 50 - 100 ms of interpreter time holding the lock while the first partition's files are being written."""
        with open(path, "rb") as f:
            data = f.read()
        if not data:
            return cls([])
        if b'"' in data or b"\r" in data or not data.endswith(b"\n"):
            return None
        a = np.frombuffer(data, np.uint8)
        ends = np.flatnonzero(a == 10)
        commas = np.flatnonzero(a == 44)
        if commas.size != ends.size:
            return None
        starts = np.concatenate([[0], ends[:-1] + 1])
        if not (np.all(commas > starts - 1) and np.all(commas < ends)):      # exactly one comma inside every line
            return None
        name_len, term_len = commas - starts, ends - commas - 1
        if name_len.max(initial=0) >= 64 or term_len.max(initial=0) >= 64:
            return None
        out = np.empty(a.size, np.uint8)
        out[1:] = a[:-1]
        out[starts] = (2 * name_len).astype(np.uint8)          # zig-zag varint of a length below 64: one byte
        out[commas + 1] = (2 * term_len).astype(np.uint8)
        self = cls.__new__(cls)
        self.count = int(ends.size)
        self.ptr = np.concatenate([starts, [a.size]]).astype(np.int64)
        self.bytes = out.tobytes()
        return self

    def __len__(self):
        return self.count

    def __getitem__(self, i):
        return self.bytes[int(self.ptr[i]):int(self.ptr[i + 1])]


def write_models_avro(path, header: bytes, sync: bytes, ids, coef_beg, coef_cnt, mean, feat_beg, feat_idx, prefix,
                      icpt_enc: bytes, class_enc: bytes, loss_enc: bytes, has_intercept: bool, threshold: float,
                      var_beg=None, variance=None, block_records=1024, deflate=False, threads=0):
    """One BayesianLinearModelAvro per entity, in the order given. prefix: list of pre-encoded string(name)+string(term)
    per global feature index, or None for an intercept-only model. Arrays are int64 / float64 numpy."""
    lib = load_library()
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    id_bytes, id_ptr = _ids_to_bytes(ids)
    keep = [np.ascontiguousarray(coef_beg, np.int64), np.ascontiguousarray(coef_cnt, np.int64),
            np.ascontiguousarray(feat_beg, np.int64), np.ascontiguousarray(mean, np.float64),
            np.ascontiguousarray(feat_idx, np.int64),
            None if var_beg is None else np.ascontiguousarray(var_beg, np.int64),
            None if variance is None else np.ascontiguousarray(variance, np.float64)]
    pre_bytes, pre_ptr = None, None
    if prefix is not None:
        prefix = EncodedFeatures.of(prefix)
        pre_ptr, pre_bytes = prefix.ptr, prefix.bytes
    t = _ModelTable(len(ids), _ptr(id_ptr), id_bytes, _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[5]), _ptr(keep[2]), _ptr(keep[3]),
                    _ptr(keep[6]), _ptr(keep[4]), _ptr(pre_ptr), pre_bytes, 0 if prefix is None else len(prefix),
                    icpt_enc, len(icpt_enc), class_enc, len(class_enc), loss_enc, len(loss_enc), int(bool(has_intercept)),
                    float(threshold))
    rc = lib.gdmix_io_avro_write_models(path.encode("utf-8"), header, len(header), sync, C.byref(t), int(block_records),
                                        int(bool(deflate)), _threads(threads))
    if rc != 0:
        raise GdmixIoError("gdmix_io_avro_write_models: " + lib.gdmix_io_last_error().decode("utf-8", "replace"))
    return len(ids)


def write_scores_avro(path, header: bytes, sync: bytes, uid, score, label, weight, per_coord, block_records=1024,
                      deflate=False, threads=0):
    lib = load_library()
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    uid = np.ascontiguousarray(uid, np.int64)
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    score, label, weight, per_coord = f32(score), f32(label), f32(weight), f32(per_coord)
    rc = lib.gdmix_io_avro_write_scores(path.encode("utf-8"), header, len(header), sync, len(uid), _ptr(uid), _ptr(score),
                                        _ptr(label), _ptr(weight), _ptr(per_coord), int(block_records), int(bool(deflate)),
                                        _threads(threads))
    if rc != 0:
        raise GdmixIoError("gdmix_io_avro_write_scores: " + lib.gdmix_io_last_error().decode("utf-8", "replace"))
    return len(uid)


def write_grouped_file(path, batch: RawBatch, entity_name, feature_bag, offset_column_name="offset", uid_column_name="uid",
                       label_column_name="response", weight_column_name="weight", int_entity_ids=False):
    """One SequenceExample per entity of `batch` into `path` (suffix .gz / .deflate => compressed)."""
    lib = load_library()
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    id_bytes, id_ptr = _ids_to_bytes(batch.entity_ids)
    keep = dict(erp=np.ascontiguousarray(batch.ent_row_ptr, np.int64), rnp=np.ascontiguousarray(batch.row_nnz_ptr, np.int64),
                col=np.ascontiguousarray(batch.col_global, np.int64), val=np.ascontiguousarray(batch.val, np.float32),
                y=np.ascontiguousarray(batch.y, np.float32), off=np.ascontiguousarray(batch.offset, np.float32),
                w=None if batch.weight is None else np.ascontiguousarray(batch.weight, np.float32),
                uid=np.ascontiguousarray(batch.uid, np.int64), idp=id_ptr)
    idbuf = C.create_string_buffer(id_bytes, len(id_bytes) + 1)
    cast = lambda a, t: None if a is None else C.cast(a.ctypes.data, C.POINTER(t))
    b = _Batch(batch.E, batch.N, batch.Z, cast(keep["erp"], C.c_int64), cast(keep["rnp"], C.c_int64), cast(keep["col"], C.c_int64),
               cast(keep["val"], C.c_float), cast(keep["y"], C.c_float), cast(keep["off"], C.c_float), cast(keep["w"], C.c_float),
               cast(keep["uid"], C.c_int64), cast(keep["idp"], C.c_int64), C.cast(idbuf, C.POINTER(C.c_char)),
               int(bool(batch.has_label)), 0)
    sc = _Schema(_enc(entity_name), _enc(feature_bag), _enc(offset_column_name), _enc(uid_column_name),
                 _enc(label_column_name), _enc(weight_column_name), -1, 0, 0)
    rc = lib.gdmix_io_write_grouped(path.encode("utf-8"), C.byref(b), C.byref(sc), int(bool(int_entity_ids)))
    if rc != 0:
        raise GdmixIoError("gdmix_io_write_grouped: " + lib.gdmix_io_last_error().decode("utf-8", "replace"))


def read_example_files(files, feature_bag, num_features, uid_name, label_name=None, offset_name=None, weight_name=None,
                       check_crc=False, threads=0):
    """Per-record (tf.train.Example) files -> dict of flat sample arrays (the fixed-effect stage's input)."""
    lib = load_library()
    sc = _Schema(None, _enc(feature_bag), _enc(offset_name), _enc(uid_name), _enc(label_name), _enc(weight_name),
                 -1 if num_features is None or feature_bag is None else int(num_features), int(bool(check_crc)), _threads(threads))
    arr = (C.c_char_p * len(files))(*[f.encode("utf-8") for f in files])
    out = C.POINTER(_Batch)()
    rc = lib.gdmix_io_read_examples(arr, len(files), C.byref(sc), C.byref(out))
    if rc != 0:
        msg = lib.gdmix_io_last_error().decode("utf-8", "replace")
        raise (ValueError if rc in (-3, -4) else GdmixIoError)(f"gdmix_io_read_examples: {msg}")
    owner = _BatchOwner(lib, out)     # the arrays are views of the library's buffers, freed with the last of them
    b = out.contents
    N, Z = int(b.N), int(b.Z)
    v = lambda ptr, n, dt: _view(owner, ptr, n, dt)
    return dict(n=N, row_nnz_ptr=v(b.row_nnz_ptr, N + 1, np.int64), col=v(b.col_global, Z, np.int64),
                val=v(b.val, Z, np.float32), y=v(b.y, N, np.float32), offset=v(b.offset, N, np.float32),
                weight=v(b.weight, N, np.float32), uid=v(b.uid, N, np.int64))


def read_models_avro(path, data_offset: int, sync: bytes, deflate: bool, prefix, icpt_enc: bytes, has_intercept: bool, threads=0):
    """Every model record of an Avro container file -> dict(ids, coef_ptr, mean, variance|None, feat_idx (the
    non-intercept coefficients' global indices, in order), has_variance). prefix: pre-encoded string(name)+string(term) per global feature index. Raises KeyError for
    a coefficient that is not in the feature list, AssertionError for a misplaced intercept (the reference's errors)."""
    lib = load_library()
    prefix = EncodedFeatures.of(prefix)
    pre_ptr, pre_bytes = prefix.ptr, prefix.bytes
    out = C.POINTER(_Models)()
    rc = lib.gdmix_io_avro_read_models(path.encode("utf-8"), int(data_offset), sync, int(bool(deflate)), _ptr(pre_ptr), pre_bytes,
                                       len(prefix), icpt_enc, len(icpt_enc), int(bool(has_intercept)), _threads(threads), C.byref(out))
    if rc != 0:
        msg = lib.gdmix_io_last_error().decode("utf-8", "replace")
        if rc == -4:
            raise (KeyError if "feature file" in msg else AssertionError)(msg)
        raise (ValueError if rc == -3 else GdmixIoError)(f"gdmix_io_avro_read_models: {msg}")
    owner = _BatchOwner(lib, out, lib.gdmix_io_free_models)
    m = out.contents
    E, Cn, Fn = int(m.E), int(m.C), int(m.F)
    id_ptr = _copy(m.id_ptr, E + 1, np.int64)
    raw = C.string_at(m.id_bytes, int(id_ptr[-1])) if E else b""
    v = lambda ptr, n, dt: _view(owner, ptr, n, dt)
    return dict(ids=EntityIds(raw, id_ptr), coef_ptr=v(m.coef_ptr, E + 1, np.int64), mean=v(m.mean, Cn, np.float64),
                variance=v(m.variance, Cn, np.float64) if m.any_variance else None,
                feat_idx=v(m.feat_idx, Fn, np.int64), has_variance=_copy(m.has_variance, E, np.uint8))


def map_coefficients(theta, cur_ptr, cur_idx, src_row, prior_coef_ptr, prior_feat_ptr, prior_theta, prior_idx, has_intercept, threads=0,
                     zero_first=False):
    """theta (float64, zero where nothing is known, laid out [intercept,] features per entity) gets the coefficients of the
    models in rows src_row (>= 0) of one table chunk; see gdmix_io_map_coefficients."""
    lib = load_library()
    i64 = lambda a: np.ascontiguousarray(a, np.int64)
    cur_ptr, cur_idx, src_row, pcp, pfp, pidx = i64(cur_ptr), i64(cur_idx), i64(src_row), i64(prior_coef_ptr), i64(prior_feat_ptr), i64(prior_idx)
    pth = np.ascontiguousarray(prior_theta, np.float64)
    assert theta.dtype == np.float64 and theta.flags.c_contiguous
    rc = lib.gdmix_io_map_coefficients(len(src_row), _ptr(cur_ptr), _ptr(cur_idx), _ptr(src_row), _ptr(pcp), _ptr(pfp), _ptr(pth),
                                       _ptr(pidx), int(bool(has_intercept)), _ptr(theta), int(bool(zero_first)), _threads(threads))
    if rc != 0:
        raise GdmixIoError("gdmix_io_map_coefficients: " + lib.gdmix_io_last_error().decode("utf-8", "replace"))
