"""Host-side statement of what gdmix-data does to a flat, per-sample dataset before the random-effect
trainer sees it, without Spark:

  * OffsetUpdater.updateOffset      gdmix-data/src/main/scala/com/linkedin/gdmix/data/OffsetUpdater.scala:105-129
        offset := predictionScore of the previous coordinate (cast to float), minus this coordinate's own
        predictionScorePerCoordinate of the previous iteration when given; inner joins on uid.
  * DataPartitioner.getGroupId      .../DataPartitioner.scala:322-380
        samples per entity are counted; with an upper bound the entity is cut into count / upper + 1 groups by
        pmod(uid, groups); with a lower bound entities below it get group -1; group 0 is the ACTIVE data
        (trained on), every other group PASSIVE data (only scored).
  * boundAndGroupData               .../DataPartitioner.scala:276-300
        one record per (entity, group): every other column collected into a per-sample list.
  * PartitionUtils.getPartitionIdUDF .../utils/PartitionUtils.scala:31-37
        partitionId = abs(javaHash(entity id as string)) % numPartitions  (contract B4, bit-exact).
  * groupPartitionAndSaveDataset    .../DataPartitioner.scala:203-274
        <out>/active/partitionId=K/..., <out>/passive/partitionId=K/... (training), <out>/partitionId=K/... otherwise.

Spark's collect_list does not promise an order inside a group; here samples keep their input order, and groups
are written in order of first appearance. Output files are entity-grouped TFRecords (SequenceExample), the
format gdmix_amd/io/grouped_reader.py and libgdmix_io.so read.
"""
import os

import numpy as np

from .batch import RawBatch, _ranges
from .io.grouped_reader import write_grouped_partition

ACTIVE, PASSIVE = "active", "passive"


# ---- Java String.hashCode on decimal / utf-8 ids ------------------------------------------------------------
def java_string_hash(s: str) -> int:
    """String.hashCode over UTF-16 code units, wrapping int32."""
    h = 0
    for cu in np.frombuffer(s.encode("utf-16-le"), np.uint16):
        h = (31 * h + int(cu)) & 0xFFFFFFFF
    return h - (1 << 32) if h >= (1 << 31) else h


def java_partition_id(entity_id, num_partitions: int) -> int:
    """Math.abs(hash) % n with Java semantics: abs(Int.MinValue) stays negative, % keeps the dividend's sign."""
    h = java_string_hash(str(entity_id))
    a = h if h == -(1 << 31) else abs(h)
    r = abs(a) % num_partitions
    return -r if a < 0 else r


def java_partition_ids_int64(ids, num_partitions: int) -> np.ndarray:
    """Vectorised java_partition_id for int64 ids rendered as decimal strings (Long.toString)."""
    v = np.asarray(ids, np.int64)
    neg = v < 0
    mag = np.where(neg, (-(v + 1)).astype(np.uint64) + np.uint64(1), v.astype(np.uint64))
    nd = np.ones(v.shape, np.int64)
    t = mag.copy()
    for _ in range(19):
        t = t // np.uint64(10)
        nd += (t > 0)
    h = np.where(neg, np.uint32(ord("-")), np.uint32(0)).astype(np.uint32)
    pow10 = np.uint64(10) ** np.arange(20, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for pos in range(19, -1, -1):   # most significant digit first
            has = nd > pos
            digit = ((mag // pow10[pos]) % np.uint64(10)).astype(np.uint32)
            h = np.where(has, h * np.uint32(31) + np.uint32(48) + digit, h)
    hs = h.astype(np.int64)
    hs = np.where(hs >= (1 << 31), hs - (1 << 32), hs)
    a = np.where(hs == -(1 << 31), hs, np.abs(hs))
    r = np.abs(a) % num_partitions
    return np.where(a < 0, -r, r).astype(np.int32)


def partition_ids(entity_ids, num_partitions: int) -> np.ndarray:
    arr = np.asarray(entity_ids)
    if arr.dtype.kind in "iu":
        return java_partition_ids_int64(arr, num_partitions)
    return np.array([java_partition_id(x, num_partitions) for x in entity_ids], np.int32)


# ---- OffsetUpdater ------------------------------------------------------------------------------------------------
def update_offsets(uid, last_uid, last_score, per_coord_uid=None, per_coord_score=None):
    """-> (rows of `uid` that survive the inner joins, their new float32 offsets)."""
    uid = np.asarray(uid, np.int64)
    last_uid = np.asarray(last_uid, np.int64)
    off = np.asarray(last_score).astype(np.float32)
    if per_coord_uid is not None:
        pu = np.asarray(per_coord_uid, np.int64)
        order = np.argsort(pu, kind="stable")
        pos = np.searchsorted(pu[order], last_uid)
        ok = (pos < pu.size) & (pu[order][np.minimum(pos, max(pu.size - 1, 0))] == last_uid) if pu.size else np.zeros(last_uid.size, bool)
        per = np.asarray(per_coord_score).astype(np.float32)
        last_uid = last_uid[ok]
        off = (off[ok] - per[order][pos[ok]]).astype(np.float32)   # FLOAT - FLOAT in Spark is float arithmetic
    order = np.argsort(last_uid, kind="stable")
    su = last_uid[order]
    pos = np.searchsorted(su, uid)
    ok = (pos < su.size) & (su[np.minimum(pos, max(su.size - 1, 0))] == uid) if su.size else np.zeros(uid.size, bool)
    rows = np.flatnonzero(ok)
    return rows, off[order][pos[rows]]


# ---- bounding and grouping --------------------------------------------------------------------------------------------
def group_ids(entity, uid, lower_bound=None, upper_bound=None) -> np.ndarray:
    """DataPartitioner.getGroupId: 0 = active, -1 = below the lower bound, > 0 = overflow groups of the upper bound."""
    entity = np.asarray(entity)
    uid = np.asarray(uid, np.int64)
    if lower_bound is None and upper_bound is None:
        return np.zeros(uid.size, np.int32)
    _, inv, cnt = np.unique(entity, return_inverse=True, return_counts=True)
    count = cnt[inv].astype(np.int64)
    # (count / upperBound + 1).cast(IntegerType): double division, truncation
    groups = (count / float(upper_bound) + 1.0).astype(np.int64) if upper_bound is not None else np.ones(uid.size, np.int64)
    gid = np.mod(uid, groups)   # pmod: non-negative for a positive modulus, as numpy's mod
    if lower_bound is not None:
        gid = np.where(count < lower_bound, -1, gid)
    return gid.astype(np.int32)


def group_samples(entity, gid):
    """Samples -> records, one per (entity, group) in order of first appearance, samples in input order.
    Returns (sample order, record start offsets [R+1], record entity, record group)."""
    entity = np.asarray(entity)
    gid = np.asarray(gid, np.int64)
    _, ent_code = np.unique(entity, return_inverse=True)
    key = ent_code.astype(np.int64) * (int(gid.max(initial=0)) - int(gid.min(initial=0)) + 1) + (gid - int(gid.min(initial=0)))
    _, first, inv = np.unique(key, return_index=True, return_inverse=True)
    rank_of_key = np.argsort(np.argsort(first, kind="stable"), kind="stable")   # key index -> order of first appearance
    rec = rank_of_key[inv]
    order = np.argsort(rec, kind="stable")
    counts = np.bincount(rec, minlength=first.size)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    starts = order[ptr[:-1]] if first.size else np.zeros(0, np.int64)
    return order, ptr, entity[starts], gid[starts]


def build_batches(entity, uid, label, offset, weight, row_nnz_ptr, col_global, val, num_partitions,
                  lower_bound=None, upper_bound=None, split=True):
    """Flat per-sample arrays (sparse bag as CSR over samples) -> {(subdir, partition id): RawBatch}.
    subdir is 'active' / 'passive' when split (training data), '' otherwise (validation data: all groups)."""
    entity = np.asarray(entity)
    uid = np.asarray(uid, np.int64)
    gid = group_ids(entity, uid, lower_bound, upper_bound)
    order, ptr, rec_entity, rec_gid = group_samples(entity, gid)
    pid = partition_ids(rec_entity, num_partitions)
    rnp = np.asarray(row_nnz_ptr, np.int64)
    k = np.diff(rnp)
    out = {}
    if split:
        sub = np.where(rec_gid == 0, 0, 1)
    else:
        sub = np.zeros(rec_gid.size, np.int64)
    for s in np.unique(sub):
        name = (ACTIVE if s == 0 else PASSIVE) if split else ""
        if split and s == 1 and lower_bound is None and upper_bound is None:
            continue
        for p in np.unique(pid[sub == s]):
            recs = np.flatnonzero((sub == s) & (pid == p))
            n = (ptr[recs + 1] - ptr[recs]).astype(np.int64)
            rows = order[_ranges(ptr[recs], n)]
            kk = k[rows]
            nz = _ranges(rnp[rows], kk)
            out[(name, int(p))] = RawBatch(
                ent_row_ptr=np.concatenate([[0], np.cumsum(n)]).astype(np.int64),
                row_nnz_ptr=np.concatenate([[0], np.cumsum(kk)]).astype(np.int64),
                col_global=np.asarray(col_global, np.int64)[nz], val=np.asarray(val, np.float32)[nz],
                y=np.asarray(label, np.float32)[rows], offset=np.asarray(offset, np.float32)[rows],
                weight=None if weight is None else np.asarray(weight, np.float32)[rows], uid=uid[rows],
                entity_ids=[str(x) for x in rec_entity[recs]], has_label=True)
    return out


def write_partitions(out_dir, batches, entity_name, feature_bag, int_entity_ids=False, suffix=".tfrecord", **names):
    """{(subdir, partition id): RawBatch} -> <out_dir>/<subdir>/partitionId=K/part-00000<suffix>; returns the paths."""
    paths = []
    for (sub, p), b in sorted(batches.items()):
        d = os.path.join(out_dir, sub, f"partitionId={p}") if sub else os.path.join(out_dir, f"partitionId={p}")
        path = os.path.join(d, "part-00000" + suffix)
        write_grouped_partition(path, b, entity_name, feature_bag, int_entity_ids=int_entity_ids, **names)
        paths.append(path)
    return paths
