"""Entity-grouped TFRecord partitions -> RawBatch.

Replaces per_entity_grouped_input_fn + dataset_reader (gdmix-trainer/src/gdmix/io/input_data_pipeline.py:
223-332, util/io_utils.py:351-364) and the per-entity slicing of prepare_jobs
(models/custom/scipy/job_consumers.py:209-258): one SequenceExample per entity with the entity id as a
scalar in the context, every dense column as a per-sample list in the context and the sparse feature bag
as `<bag>_indices` / `<bag>_values` feature lists (one step per sample).
"""
import glob
import os

import numpy as np

from .. import constants
from ..batch import RawBatch
from . import tfrecord
from .metadata import DatasetMetadata

INDICES_SUFFIX = "_indices"
VALUES_SUFFIX = "_values"


def resolve_input_files(input_path):
    """Directory -> sorted *.tfrecord, else *.tfrecord.deflate, else *.tfrecord.gz; a file or pattern is
    used as given (input_data_pipeline.py:88-126; files are sorted, distribution_utils.py:36)."""
    if os.path.isdir(input_path):
        base = os.path.join(input_path, constants.TFRECORD_GLOB_PATTERN)
        for pattern in (base, base + ".deflate", base + ".gz"):
            files = sorted(glob.glob(pattern))
            if files:
                return files
        return []
    return sorted(glob.glob(input_path)) if any(c in input_path for c in "*?[") else [input_path]


def _column(ctx, name, kinds, entity_label):
    if name not in ctx:
        raise KeyError(f"column {name!r} is missing from the record of entity {entity_label}")
    kind, vals = ctx[name]
    if kind == "empty":
        return np.zeros(0, np.float32)
    if kind not in kinds:
        raise ValueError(f"column {name!r} of entity {entity_label} has kind {kind}, expected one of {kinds}")
    return vals


def read_grouped_partition(input_path, metadata, entity_name, feature_bag, offset_column_name,
                           uid_column_name, label_column_name=None, weight_column_name=None,
                           num_features=None, check_crc=False, native=None, threads=0, wire=False):
    """Read every record under input_path into one RawBatch (entity order = file order, then record order).

    feature_bag None => intercept-only model: one dummy zero feature per sample (job_consumers.py:213-218).
    label_column_name None (or absent from the records) => batch.has_label False (inference data).
    native: True = libgdmix_io.so (multi-threaded C++), False = the Python decoder below, None = native when the
    library has been built. Both follow the same rules (tests/test_native_io.py).
    wire: with the native reader, hand the partition over in the 32-bit form (batch.WireRawBatch); ignored by the Python decoder.
    """
    md = metadata if isinstance(metadata, DatasetMetadata) else DatasetMetadata(metadata)
    if entity_name not in md.get_feature_names():
        raise ValueError(f"entity name {entity_name} is not found among the features")
    has_weight_col = weight_column_name is not None and weight_column_name in md.get_feature_names()
    files = resolve_input_files(input_path)
    from . import native_reader
    if native is None:
        native = native_reader.available()
    if native:
        return native_reader.read_grouped_files(files, entity_name, feature_bag, offset_column_name, uid_column_name,
                                                label_column_name, weight_column_name if has_weight_col else None,
                                                num_features, check_crc, threads, wire=wire)
    ent_n, row_k = [], []
    cols, vals, ys, offs, ws, uids, ids = [], [], [], [], [], [], []
    has_label = label_column_name is not None
    for fn in files:
        for rec in tfrecord.iter_records(fn, check_crc=check_crc):
            ctx, fls = tfrecord.decode_sequence_example(rec)
            if entity_name not in ctx:
                raise KeyError(f"entity column {entity_name!r} is missing from a record of {fn}")
            kind, ev = ctx[entity_name]
            if len(ev) != 1:
                raise ValueError(f"entity column {entity_name!r} must be a scalar, got {len(ev)} values")
            # job_consumers.py:235-239: bytes -> utf-8, everything else -> str()
            eid = ev[0].decode("utf-8") if kind == "bytes" else str(int(ev[0]))
            uid = _column(ctx, uid_column_name, ("int64",), eid)
            n = len(uid)
            off = _column(ctx, offset_column_name, ("float",), eid)
            if len(off) != n:
                raise ValueError(f"entity {eid}: {len(off)} offsets for {n} uids")
            if feature_bag is None:
                if num_features not in (None, 1):
                    raise AssertionError(f"an intercept-only model has one dummy feature, not {num_features}")
                k = np.ones(n, np.int64)
                c = np.zeros(n, np.int64)
                v = np.zeros(n, np.float32)
            else:
                istep = fls.get(feature_bag + INDICES_SUFFIX, [])
                vstep = fls.get(feature_bag + VALUES_SUFFIX, [])
                if len(istep) != len(vstep):
                    raise ValueError(f"entity {eid}: {len(istep)} index lists vs {len(vstep)} value lists")
                k = np.array([len(s[1]) for s in istep], np.int64)
                kv = np.array([len(s[1]) for s in vstep], np.int64)
                if not np.array_equal(k, kv):
                    raise ValueError(f"entity {eid}: indices and values of {feature_bag} differ in length")
                # the reference derives the sample count from the last sample that owns a feature and
                # asserts it equals the uid count (job_consumers.py:229-232)
                nz = np.flatnonzero(k)
                sample_count = int(nz[-1]) + 1 if nz.size else 0
                if sample_count != n:   # an explicit raise: `python -O` must not turn this data check off
                    raise AssertionError(f"entity {eid}: {sample_count} feature rows (last non-empty) vs {n} uids")
                k = k[:n]
                c = np.concatenate([np.asarray(s[1], np.int64) for s in istep[:n]]) if n else np.zeros(0, np.int64)
                v = np.concatenate([np.asarray(s[1], np.float32) for s in vstep[:n]]) if n else np.zeros(0, np.float32)
                if num_features is not None and c.size and (c.min() < 0 or c.max() >= num_features):
                    raise ValueError(f"entity {eid}: feature index outside [0, {num_features})")
            if has_label and label_column_name in ctx:
                lab = _column(ctx, label_column_name, ("int64", "float"), eid)
                if len(lab) != n:
                    raise ValueError(f"entity {eid}: {len(lab)} labels for {n} uids")
                ys.append(np.asarray(lab, np.float32))
            else:
                has_label = False
                ys.append(np.zeros(n, np.float32))
            if has_weight_col:
                w = _column(ctx, weight_column_name, ("float",), eid)
                if len(w) != n:
                    raise ValueError(f"entity {eid}: {len(w)} weights for {n} uids")
                ws.append(np.asarray(w, np.float32))
            ent_n.append(n)
            row_k.append(k)
            cols.append(c)
            vals.append(v)
            offs.append(np.asarray(off, np.float32))
            uids.append(np.asarray(uid, np.int64))
            ids.append(eid)

    def cat(parts, dt):
        return np.concatenate(parts).astype(dt) if parts else np.zeros(0, dt)
    rk = cat(row_k, np.int64)
    return RawBatch(ent_row_ptr=np.concatenate([[0], np.cumsum(np.array(ent_n, np.int64))]).astype(np.int64),
                    row_nnz_ptr=np.concatenate([[0], np.cumsum(rk)]).astype(np.int64),
                    col_global=cat(cols, np.int64), val=cat(vals, np.float32), y=cat(ys, np.float32),
                    offset=cat(offs, np.float32), weight=cat(ws, np.float32) if has_weight_col else None,
                    uid=cat(uids, np.int64), entity_ids=ids, has_label=has_label)


def write_grouped_partition(path, batch: RawBatch, entity_name, feature_bag, offset_column_name="offset",
                            uid_column_name="uid", label_column_name="response", weight_column_name="weight",
                            int_entity_ids=False, native=None):
    """Write a RawBatch as one SequenceExample per entity (layout of DataPartitioner's output,
    SURVEY.md Appendix A). Used by tests, the partitioner tool and to materialise synthetic partitions.
    native None => libgdmix_io.so when built (same bytes for uncompressed files)."""
    from . import native_reader
    if native is None:
        native = native_reader.available()
    if native:
        native_reader.write_grouped_file(path, batch, entity_name, feature_bag, offset_column_name, uid_column_name,
                                         label_column_name, weight_column_name, int_entity_ids)
        return
    payloads = []
    n = batch.ent_n()
    for e in range(batch.E):
        r0 = int(batch.ent_row_ptr[e])
        sl = slice(r0, r0 + int(n[e]))
        eid = batch.entity_ids[e]
        ctx = {entity_name: ("int64", [int(eid)]) if int_entity_ids else ("bytes", [eid.encode("utf-8")]),
               uid_column_name: ("int64", batch.uid[sl]),
               offset_column_name: ("float", batch.offset[sl])}
        if batch.has_label and label_column_name:
            ctx[label_column_name] = ("int64", batch.y[sl].astype(np.int64))
        if batch.weight is not None and weight_column_name:
            ctx[weight_column_name] = ("float", batch.weight[sl])
        fls = {}
        if feature_bag is not None:
            isteps, vsteps = [], []
            for i in range(r0, r0 + int(n[e])):
                z0, z1 = int(batch.row_nnz_ptr[i]), int(batch.row_nnz_ptr[i + 1])
                isteps.append(("int64", batch.col_global[z0:z1]))
                vsteps.append(("float", batch.val[z0:z1]))
            fls[feature_bag + INDICES_SUFFIX] = isteps
            fls[feature_bag + VALUES_SUFFIX] = vsteps
        payloads.append(tfrecord.encode_sequence_example(ctx, fls))
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    tfrecord.write_records(path, payloads)
