"""Host-side containers for an entity-grouped random-effect batch.

A RawBatch is the flattened form of what the reference's prepare_jobs slices one entity at a time out
of the TF sparse tensors (gdmix-trainer/src/gdmix/models/custom/scipy/job_consumers.py:176-258):
entity-major, then sample-major, then non-zero-major ragged arrays.
"""
from dataclasses import dataclass
from typing import List, Optional

import numpy as np


@dataclass
class RawBatch:
    ent_row_ptr: np.ndarray            # int64 [E+1] sample offsets per entity
    row_nnz_ptr: np.ndarray            # int64 [N+1] non-zero offsets per sample
    col_global: np.ndarray             # int64 [Z]   global feature index
    val: np.ndarray                    # float32 [Z]
    y: np.ndarray                      # float32 [N] 0/1
    offset: np.ndarray                 # float32 [N]
    weight: Optional[np.ndarray] = None  # float32 [N] or None (=> ones)
    uid: Optional[np.ndarray] = None     # int64 [N] sample ids (scoring output only)
    entity_ids: Optional[List[str]] = None  # [E] str(entity id) as job_consumers.py:235-239 renders it
    has_label: bool = True
    binary_labels: bool = True              # False: real-valued labels (fixed-effect linear regression)
    trusted: bool = False                   # built by the native reader, which checked all of validate() while decoding

    def __post_init__(self):
        self.ent_row_ptr = np.ascontiguousarray(self.ent_row_ptr, np.int64)
        self.row_nnz_ptr = np.ascontiguousarray(self.row_nnz_ptr, np.int64)
        self.col_global = np.ascontiguousarray(self.col_global, np.int64)
        self.val = np.ascontiguousarray(self.val, np.float32)
        self.y = np.ascontiguousarray(self.y, np.float32)
        self.offset = np.ascontiguousarray(self.offset, np.float32)
        if self.weight is not None:
            self.weight = np.ascontiguousarray(self.weight, np.float32)
        if self.uid is not None:
            self.uid = np.ascontiguousarray(self.uid, np.int64)
        if not self.trusted:
            self.validate()

    @property
    def E(self):
        return self.ent_row_ptr.size - 1

    @property
    def N(self):
        return self.row_nnz_ptr.size - 1

    @property
    def Z(self):
        return self.col_global.size

    def validate(self):
        E, N, Z = self.E, self.N, self.Z
        if E < 0 or N < 0:
            raise ValueError("ent_row_ptr / row_nnz_ptr need at least one element")
        if self.ent_row_ptr[0] != 0 or self.ent_row_ptr[-1] != N:
            raise ValueError("ent_row_ptr must run from 0 to N")
        if self.row_nnz_ptr[0] != 0 or self.row_nnz_ptr[-1] != Z:
            raise ValueError("row_nnz_ptr must run from 0 to Z")
        if np.any(np.diff(self.ent_row_ptr) < 0) or np.any(np.diff(self.row_nnz_ptr) < 0):
            raise ValueError("offset arrays must be non-decreasing")
        for name in ("y", "offset"):
            if getattr(self, name).size != N:
                raise ValueError(f"{name} must have N={N} elements")
        if self.val.size != Z:
            raise ValueError(f"val must have Z={Z} elements")
        if self.weight is not None and self.weight.size != N:
            raise ValueError("weight must have N elements")
        if self.uid is not None and self.uid.size != N:
            raise ValueError("uid must have N elements")
        if self.entity_ids is not None and len(self.entity_ids) != E:
            raise ValueError("entity_ids must have E elements")
        # fit() asserts labels are 0/1 (binary_logistic_regression.py:208)
        if self.has_label and self.binary_labels and N and np.count_nonzero((self.y != 0) & (self.y != 1)):
            raise AssertionError("labels must be 0 or 1")

    def ent_nnz(self):
        return np.diff(self.row_nnz_ptr[self.ent_row_ptr])

    def ent_n(self):
        return np.diff(self.ent_row_ptr)

    def to_wire(self):
        """The 32-bit hand-over form (include/gdmix_re.h, gdmix_re_wire_batch): counts instead of int64 pointers, int32 feature
        ids, byte labels. dict of contiguous numpy arrays + the two widths; 0.6 of the raw form's bytes for C2."""
        if self.Z and (self.col_global.min() < 0 or self.col_global.max() > 0x7fffffff):
            raise ValueError("feature index outside [0, 2^31)")
        ent_n = np.diff(self.ent_row_ptr)
        if self.E and ent_n.max() > 0x7fffffff:
            raise ValueError("an entity has more than 2^31 samples")
        k = np.diff(self.row_nnz_ptr)
        kmax = int(k.max()) if k.size else 0
        kdt = np.uint8 if kmax <= 0xff else (np.uint16 if kmax <= 0xffff else np.uint32)
        binary = self.binary_labels and self.has_label
        cdt = np.uint16 if (self.Z == 0 or self.col_global.max() <= 0xffff) else np.int32
        return dict(E=self.E, N=self.N, Z=self.Z, ent_n=ent_n.astype(np.int32), row_nnz=k.astype(kdt),
                    row_nnz_width=np.dtype(kdt).itemsize, col_global=self.col_global.astype(cdt), col_width=np.dtype(cdt).itemsize,
                    val=self.val,
                    y=self.y.astype(np.uint8) if binary else self.y, y_width=1 if binary else 4, offset=self.offset,
                    weight=self.weight)

    def select(self, ents):
        """Sub-batch of the given entity indices (ascending order not required)."""
        ents = np.asarray(ents, np.int64)
        n = self.ent_n()[ents]
        rows = _ranges(self.ent_row_ptr[ents], n)
        rz = np.diff(self.row_nnz_ptr)[rows]
        nz = _ranges(self.row_nnz_ptr[rows], rz)
        return RawBatch(
            ent_row_ptr=np.concatenate([[0], np.cumsum(n)]),
            row_nnz_ptr=np.concatenate([[0], np.cumsum(rz)]),
            col_global=self.col_global[nz], val=self.val[nz], y=self.y[rows], offset=self.offset[rows],
            weight=None if self.weight is None else self.weight[rows],
            uid=None if self.uid is None else self.uid[rows],
            entity_ids=None if self.entity_ids is None else [self.entity_ids[i] for i in ents],
            has_label=self.has_label)


class WireRawBatch(RawBatch):
    """A partition as libgdmix_io's gdmix_io_narrow leaves it: the 32-bit hand-over form (per-sample counts, uint16 / int32 feature
    indices, byte labels) next to the arrays both forms share (ent_row_ptr, val, y, offset, weight, uid, ids). to_wire() hands the
    library's arrays over as they are — no pass over the partition on this side; the two large 64-bit arrays of RawBatch (row_nnz_ptr,
    col_global) exist only if somebody asks for them (host-side consumers: select(), the CPU stand-ins of the tests), rebuilt then by a
    cumsum and a widening copy."""

    def __init__(self, ent_row_ptr, ent_n, row_nnz, col, val, y, y8, offset, weight=None, uid=None, entity_ids=None, has_label=True):
        d = self.__dict__
        d.update(ent_row_ptr=ent_row_ptr, val=val, y=y, offset=offset, weight=weight, uid=uid, entity_ids=entity_ids, has_label=has_label,
                 binary_labels=True, trusted=True, _ent_n=ent_n, _row_nnz=row_nnz, _col=col, _y8=y8, _row_nnz_ptr=None, _col_global=None)

    @property
    def row_nnz_ptr(self):
        if self._row_nnz_ptr is None:
            p = np.zeros(self._row_nnz.size + 1, np.int64)
            np.cumsum(self._row_nnz, dtype=np.int64, out=p[1:])
            self.__dict__["_row_nnz_ptr"] = p
        return self._row_nnz_ptr

    @property
    def col_global(self):
        if self._col_global is None:
            self.__dict__["_col_global"] = self._col.astype(np.int64)
        return self._col_global

    @property
    def N(self):
        return self._row_nnz.size

    @property
    def Z(self):
        return self._col.size

    def ent_n(self):
        return self._ent_n.astype(np.int64)

    def ent_nnz(self):
        """Non-zeros per entity straight from the narrow per-sample counts (no 64-bit row pointer array is built: ADVICE r4)."""
        if self._row_nnz_ptr is not None:
            return RawBatch.ent_nnz(self)
        E = self.E
        if E == 0 or self._row_nnz.size == 0:
            return np.zeros(E, np.int64)
        starts = np.asarray(self.ent_row_ptr[:-1], np.int64)
        live = starts < self._row_nnz.size          # (entities without samples at the end of the batch)
        out = np.zeros(E, np.int64)
        s = np.add.reduceat(self._row_nnz, starts[live], dtype=np.int64)
        s[np.diff(np.append(starts[live], self._row_nnz.size)) == 0] = 0     # reduceat returns the element itself for an empty range
        out[live] = s
        return out

    def to_wire(self):
        binary = self._y8 is not None
        return dict(E=self.E, N=self.N, Z=self.Z, ent_n=self._ent_n, row_nnz=self._row_nnz, row_nnz_width=self._row_nnz.dtype.itemsize,
                    col_global=self._col, col_width=self._col.dtype.itemsize, val=self.val, y=self._y8 if binary else self.y,
                    y_width=1 if binary else 4, offset=self.offset, weight=self.weight)


def _ranges(starts, lens):
    """Concatenate arange(s, s+l) for every (s, l) without a Python loop."""
    lens = np.asarray(lens, np.int64)
    starts = np.asarray(starts, np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, np.int64)
    out_start = np.cumsum(lens) - lens
    return np.arange(total, dtype=np.int64) - np.repeat(out_start, lens) + np.repeat(starts, lens)


def concat(batches):
    batches = list(batches)
    if not batches:
        raise ValueError("no batches")
    erp, rnp, n_off, z_off = [np.zeros(1, np.int64)], [np.zeros(1, np.int64)], 0, 0
    for b in batches:
        erp.append(b.ent_row_ptr[1:] + n_off)
        rnp.append(b.row_nnz_ptr[1:] + z_off)
        n_off += b.N
        z_off += b.Z
    any_w = any(b.weight is not None for b in batches)
    any_u = all(b.uid is not None for b in batches)
    any_id = all(b.entity_ids is not None for b in batches)
    return RawBatch(
        ent_row_ptr=np.concatenate(erp), row_nnz_ptr=np.concatenate(rnp),
        col_global=np.concatenate([b.col_global for b in batches]),
        val=np.concatenate([b.val for b in batches]), y=np.concatenate([b.y for b in batches]),
        offset=np.concatenate([b.offset for b in batches]),
        weight=np.concatenate([b.weight if b.weight is not None else np.ones(b.N, np.float32)
                               for b in batches]) if any_w else None,
        uid=np.concatenate([b.uid for b in batches]) if any_u else None,
        entity_ids=sum([list(b.entity_ids) for b in batches], []) if any_id else None,
        has_label=all(b.has_label for b in batches))
