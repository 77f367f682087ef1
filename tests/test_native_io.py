"""libgdmix_io.so (native TFRecord reader, include/gdmix_io.h) against the Python statement of the same rules
(gdmix_amd/io/grouped_reader.py) and against the reference's own fixture files. CPU only."""
import os
import re

import numpy as np
import pytest

from gdmix_amd import synthetic
from gdmix_amd.batch import RawBatch, WireRawBatch
from gdmix_amd.io import native_reader, tfrecord
from gdmix_amd.io.grouped_reader import read_grouped_partition, write_grouped_partition
from helpers import load_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = os.path.join(ROOT, "tests", "golden", "ref_resources")
ARRAYS = ("ent_row_ptr", "row_nnz_ptr", "col_global", "val", "y", "offset", "uid")

MD = {"features": [{"name": "bag", "dtype": "float", "shape": [4096], "isSparse": True},
                   {"name": "weight", "dtype": "float", "shape": [], "isSparse": False},
                   {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                   {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                   {"name": "ent", "dtype": "string", "shape": [], "isSparse": False}],
      "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}


def same(a: RawBatch, b: RawBatch):
    for k in ARRAYS:
        x, y = getattr(a, k), getattr(b, k)
        assert x.dtype == y.dtype, k
        np.testing.assert_array_equal(x, y, err_msg=k)
    assert (a.weight is None) == (b.weight is None)
    if a.weight is not None:
        np.testing.assert_array_equal(a.weight, b.weight)
    assert a.entity_ids == b.entity_ids
    assert a.has_label == b.has_label


def both(path, *args, **kw):
    py = read_grouped_partition(path, *args, native=False, **kw)
    for threads in (1, 4):
        nat = read_grouped_partition(path, *args, native=True, threads=threads, **kw)
        same(py, nat)
        same_wire(py, read_grouped_partition(path, *args, native=True, threads=threads, wire=True, **kw))
    return py


def same_wire(py: RawBatch, w):
    """The batch the library narrowed (gdmix_io_narrow): its hand-over arrays are the ones RawBatch.to_wire() derives from the 64-bit
    arrays, element for element and width for width; the 64-bit arrays it rebuilds on demand are the originals."""
    assert isinstance(w, WireRawBatch) and (w.E, w.N, w.Z) == (py.E, py.N, py.Z)
    assert w._row_nnz_ptr is None and w._col_global is None       # nothing 64-bit until somebody asks
    a, b = py.to_wire(), w.to_wire()
    assert a.keys() == b.keys()
    for k, x in a.items():
        if isinstance(x, np.ndarray):
            assert b[k].dtype == x.dtype and b[k].flags.c_contiguous, k
            np.testing.assert_array_equal(b[k], x, err_msg=k)
        else:
            assert b[k] == x or (x is None and b[k] is None), k
    np.testing.assert_array_equal(w.ent_n(), py.ent_n())
    np.testing.assert_array_equal(w.ent_nnz(), py.ent_nnz())
    same(py, w)
    if py.E > 1:
        same(py.select([py.E - 1, 0]), w.select([py.E - 1, 0]))


def test_library_exports_every_symbol_the_header_declares():
    hdr = open(os.path.join(ROOT, "include", "gdmix_io.h")).read()
    declared = set(re.findall(r"GDMIX_IO_API\s+[\w\s\*]+?\b(gdmix_io_\w+)\s*\(", hdr))
    assert declared == set(native_reader.EXPORTED_SYMBOLS)
    lib = native_reader.load_library()
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_crc32c_known_answers():
    assert native_reader.crc32c(b"123456789") == 0xE3069283
    assert native_reader.crc32c(b"") == 0
    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 63, 1000):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert native_reader.crc32c(data) == tfrecord.crc32c(data)
        assert native_reader.masked_crc32c(data) == tfrecord.masked_crc32c(data)


@pytest.mark.parametrize("meta,entity,bag,ids", [("data.json", "memberId", "per_member", ["100034", "100"]),
                                                 ("data_intercept_only.json", "memberId", None, ["100034", "100"])])
def test_reference_fixture(meta, entity, bag, ids):
    b = both(os.path.join(RES, "data.tfrecord"), os.path.join(RES, meta), entity, bag, "offset", "uid", "response", "weight",
             num_features=100 if bag else 1, check_crc=True)
    assert b.entity_ids == ids
    assert b.uid.tolist() == [10, 20, 23] and b.y.tolist() == [0, 1, 1]
    if bag:
        assert b.col_global.tolist() == [0, 7, 60, 80, 95, 34, 57, 10, 11]
    else:
        assert b.row_nnz_ptr.tolist() == [0, 1, 2, 3] and not b.val.any()


@pytest.mark.parametrize("suffix", [".tfrecord", ".tfrecord.gz", ".tfrecord.deflate"])
@pytest.mark.parametrize("int_ids", [False, True])
def test_round_trip_matches_python_reader(tmp_path, suffix, int_ids):
    b = synthetic.make_ragged_batch(700, seed=11)
    if int_ids:
        b.entity_ids = [str(7 * i - 3) for i in range(b.E)]   # includes a negative id
    # several files, file order = sorted name order
    cuts = [0, 250, 251, 700]
    for i in range(3):
        part = b.select(np.arange(cuts[i], cuts[i + 1]))
        write_grouped_partition(str(tmp_path / f"part-{i:05d}{suffix}"), part, "ent", "bag", int_entity_ids=int_ids)
    r = both(str(tmp_path), MD, "ent", "bag", "offset", "uid", "response", "weight", num_features=4096, check_crc=True)
    for k in ARRAYS:
        np.testing.assert_array_equal(getattr(r, k), getattr(b, k), err_msg=k)
    assert r.entity_ids == b.entity_ids


def test_no_weight_column_and_inference_data(tmp_path):
    b, _, _, _ = load_fixture("ragged")
    nolab = RawBatch(**{**{k: getattr(b, k) for k in ARRAYS}, "weight": None, "entity_ids": b.entity_ids, "has_label": False})
    write_grouped_partition(str(tmp_path / "a.tfrecord"), nolab, "ent", "bag")
    md = {"features": [f for f in MD["features"] if f["name"] != "weight"], "labels": MD["labels"]}
    r = both(str(tmp_path), md, "ent", "bag", "offset", "uid", "response", "weight", num_features=4096)
    assert r.weight is None and not r.has_label and not r.y.any()
    r2 = both(str(tmp_path), md, "ent", "bag", "offset", "uid", None, None)
    assert not r2.has_label


def _record(ctx, fls):
    return tfrecord.encode_sequence_example(ctx, fls)


def _write(path, payloads):
    tfrecord.write_records(str(path), payloads)
    return str(path)


def test_labels_are_dropped_from_the_first_unlabelled_record_on(tmp_path):
    def rec(eid, with_label):
        ctx = {"ent": ("bytes", [eid]), "uid": ("int64", [1, 2]), "offset": ("float", [0.5, 0.25])}
        if with_label:
            ctx["response"] = ("int64", [1, 0])
        return _record(ctx, {"bag_indices": [("int64", [3]), ("int64", [4, 5])], "bag_values": [("float", [1.0]), ("float", [2.0, 3.0])]})
    p = _write(tmp_path / "x.tfrecord", [rec(b"a", True), rec(b"b", False), rec(b"c", True)])
    r = both(p, MD, "ent", "bag", "offset", "uid", "response", None)
    assert not r.has_label and r.y.tolist() == [1, 0, 0, 0, 0, 0]


def test_unpacked_lists_float_labels_and_trailing_empty_steps(tmp_path):
    # hand-encoded record: unpacked int64 / float lists, float labels, a third (empty) step after the last sample
    def ld(fn, payload):
        return bytes([(fn << 3) | 2]) + tfrecord._enc_varint(len(payload)) + payload
    f32 = lambda x: np.float32(x).tobytes()
    int_unpacked = ld(3, b"\x08\x05\x08\x07")                          # Int64List{5, 7} one field per element
    flt_unpacked = ld(2, b"\x0d" + f32(1.5) + b"\x0d" + f32(-2.0))     # FloatList{1.5, -2.0}
    entry = lambda k, feat: ld(1, ld(1, k) + ld(2, feat))
    context = (entry(b"ent", ld(3, ld(1, tfrecord._enc_varint(42)))) + entry(b"uid", int_unpacked) +
               entry(b"offset", flt_unpacked) + entry(b"response", ld(2, ld(1, f32(1.0) + f32(0.0)))))
    step = lambda feat: ld(1, feat)
    idx = step(ld(3, b"\x08\x09")) + step(ld(3, ld(1, b"\x01\x02"))) + step(ld(3, b""))
    val = step(ld(2, b"\x0d" + f32(4.0))) + step(ld(2, ld(1, f32(5.0) + f32(6.0)))) + step(ld(2, b""))
    flists = entry(b"bag_indices", idx) + entry(b"bag_values", val)
    p = _write(tmp_path / "u.tfrecord", [ld(1, context) + ld(2, flists)])
    r = both(p, MD, "ent", "bag", "offset", "uid", "response", None, num_features=10)
    assert r.entity_ids == ["42"] and r.uid.tolist() == [5, 7] and r.y.tolist() == [1, 0]
    assert r.col_global.tolist() == [9, 1, 2] and r.val.tolist() == [4, 5, 6] and r.row_nnz_ptr.tolist() == [0, 1, 3]


def _bad_cases():
    good_ctx = {"ent": ("bytes", [b"e"]), "uid": ("int64", [1, 2]), "offset": ("float", [0.0, 0.0]), "response": ("int64", [0, 1])}
    good_fl = {"bag_indices": [("int64", [1]), ("int64", [2])], "bag_values": [("float", [1.0]), ("float", [2.0])]}
    def without(d, k):
        return {a: b for a, b in d.items() if a != k}
    yield "missing uid", without(good_ctx, "uid"), good_fl
    yield "missing offset", without(good_ctx, "offset"), good_fl
    yield "missing entity", without(good_ctx, "ent"), good_fl
    yield "two entity ids", {**good_ctx, "ent": ("bytes", [b"a", b"b"])}, good_fl
    yield "offset length", {**good_ctx, "offset": ("float", [0.0])}, good_fl
    yield "label length", {**good_ctx, "response": ("int64", [0])}, good_fl
    yield "value list length", good_ctx, {**good_fl, "bag_values": [("float", [1.0]), ("float", [2.0, 3.0])]}
    yield "step count", good_ctx, {**good_fl, "bag_values": [("float", [1.0])]}
    yield "last sample without a feature", good_ctx, {"bag_indices": [("int64", [1]), ("int64", [])], "bag_values": [("float", [1.0]), ("float", [])]}
    yield "index out of range", good_ctx, {**good_fl, "bag_indices": [("int64", [1]), ("int64", [5000])]}
    yield "uid is a float list", {**good_ctx, "uid": ("float", [1.0, 2.0])}, good_fl


@pytest.mark.parametrize("case", list(_bad_cases()), ids=lambda c: c[0])
def test_schema_violations_fail_in_both_readers(tmp_path, case):
    _, ctx, fls = case
    p = _write(tmp_path / "bad.tfrecord", [_record(ctx, fls)])
    for native in (False, True):
        with pytest.raises((ValueError, KeyError, AssertionError)):
            read_grouped_partition(p, MD, "ent", "bag", "offset", "uid", "response", None, num_features=4096, native=native)


def test_corrupt_and_truncated_files_fail(tmp_path):
    b, _, _, _ = load_fixture("ragged")
    good = str(tmp_path / "g.tfrecord")
    write_grouped_partition(good, b, "ent", "bag")
    data = bytearray(open(good, "rb").read())
    flipped = bytearray(data)
    flipped[len(data) // 2] ^= 0x40
    cases = {"crc.tfrecord": (bytes(flipped), True), "cut.tfrecord": (bytes(data[:len(data) - 7]), False),
             "hdr.tfrecord": (bytes(data[:5]), False)}
    for name, (payload, crc) in cases.items():
        p = str(tmp_path / name)
        open(p, "wb").write(payload)
        for native in (False, True):
            with pytest.raises((ValueError, native_reader.GdmixIoError)):
                read_grouped_partition(p, MD, "ent", "bag", "offset", "uid", "response", "weight", check_crc=crc, native=native)
    gz = str(tmp_path / "bad.tfrecord.gz")
    open(gz, "wb").write(b"\x1f\x8b\x08\x00garbage-not-a-gzip-stream")
    with pytest.raises((ValueError, native_reader.GdmixIoError)):
        read_grouped_partition(gz, MD, "ent", "bag", "offset", "uid", "response", "weight", native=True)
    with pytest.raises(native_reader.GdmixIoError):
        native_reader.read_grouped_files([str(tmp_path / "does-not-exist.tfrecord")], "ent", "bag", "offset", "uid")


def test_empty_inputs(tmp_path):
    open(tmp_path / "empty.tfrecord", "wb").close()
    r = both(str(tmp_path / "empty.tfrecord"), MD, "ent", "bag", "offset", "uid", "response", "weight")
    assert r.E == 0 and r.N == 0 and r.Z == 0
    r = native_reader.read_grouped_files([], "ent", "bag", "offset", "uid")
    assert r.E == 0


# ---- Avro writers: byte for byte what the Python encoders write --------------------------------------------
def _table(seed, E, D, with_var, id_base=0):
    from gdmix_amd.model import ModelTable
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 9, E)
    feat_ptr = np.concatenate([[0], np.cumsum(d)])
    coef_ptr = feat_ptr + np.arange(E + 1)
    idx = np.concatenate([np.sort(rng.choice(D, k, replace=False)) for k in d]) if d.sum() else np.zeros(0, np.int64)
    theta = rng.standard_normal(coef_ptr[-1])
    theta[rng.random(theta.size) < 0.3] *= 1e-5          # below the export threshold
    theta[rng.random(theta.size) < 0.05] = 0.0
    var = rng.random(theta.size) if with_var else None
    t = ModelTable()
    t.add_chunk([f"id{id_base + i}" if i % 3 else str(id_base + i) for i in range(E)], theta, coef_ptr, idx, feat_ptr, var)
    return t


@pytest.mark.parametrize("with_variance", [False, True])
def test_native_model_writer_is_byte_identical(tmp_path, with_variance):
    from gdmix_amd.model import _export_models_to_avro
    D = 50
    feature_list = [(f"name{j}", "" if j % 4 else f"term{j}") for j in range(D)]
    table = _table(1, 2500, D, with_variance)
    # a second chunk overriding some ids (dict.update order: existing ids keep their place), without variance
    table.update(_table(2, 300, D, False, id_base=2400))
    sync = bytes(range(16))
    a, b = str(tmp_path / "py.avro"), str(tmp_path / "native.avro")
    na = _export_models_to_avro(a, table, feature_list, True, with_variance, native=False, sync_marker=sync)
    nb = _export_models_to_avro(b, table, feature_list, True, with_variance, native=True, sync_marker=sync)
    assert na == nb == len(table) == 2700
    assert open(a, "rb").read() == open(b, "rb").read()
    # and the file reads back through the generic codec
    from gdmix_amd.io import avro
    recs = list(avro.read_file(b))
    assert len(recs) == 2700 and recs[0]["modelId"] == "0" and recs[0]["means"][0]["name"] == "(INTERCEPT)"


def test_native_model_writer_intercept_only_and_no_intercept(tmp_path):
    from gdmix_amd.model import ModelTable, _export_models_to_avro
    sync = b"s" * 16
    t = ModelTable()
    t.add_chunk(["a", "b", "c"], np.array([0.5, -2.0, 1e-9]), [0, 1, 2, 3], np.zeros(0, np.int64), [0, 0, 0, 0])
    a, b = str(tmp_path / "py.avro"), str(tmp_path / "nat.avro")
    _export_models_to_avro(a, t, None, True, False, native=False, sync_marker=sync)
    _export_models_to_avro(b, t, None, True, False, native=True, sync_marker=sync)
    assert open(a, "rb").read() == open(b, "rb").read()
    t2 = ModelTable()
    t2.add_chunk(["x", "y"], np.array([0.5, -2.0, 3.0]), [0, 2, 3], np.array([1, 0, 1]), [0, 2, 3])
    fl = [("f0", ""), ("f1", "t")]
    _export_models_to_avro(a, t2, fl, False, False, native=False, sync_marker=sync)
    _export_models_to_avro(b, t2, fl, False, False, native=True, sync_marker=sync)
    assert open(a, "rb").read() == open(b, "rb").read()
    with pytest.raises(native_reader.GdmixIoError):   # a feature index outside the feature list
        _export_models_to_avro(b, t2, fl[:1], False, False, native=True, sync_marker=sync)


@pytest.mark.parametrize("has_label,has_weight", [(True, True), (False, True), (True, False), (False, False)])
def test_native_score_writer_is_byte_identical(tmp_path, has_label, has_weight):
    from types import SimpleNamespace
    from gdmix_amd.io import avro
    from gdmix_amd.model import _write_scores
    sp = SimpleNamespace(uid_column_name="uid", prediction_score_column_name="predictionScore", label_column_name="response",
                         weight_column_name="weight", prediction_score_per_coordinate_column_name="predictionScorePerCoordinate")
    schema = avro.inference_output_schema(sp, has_weight=has_weight)
    rng = np.random.default_rng(3)
    n = 5000
    uid = rng.integers(-2 ** 62, 2 ** 62, n)
    score, per = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    label = (rng.random(n) < 0.5).astype(np.float32) if has_label else None
    weight = rng.random(n).astype(np.float32) if has_weight else None
    sync = bytes(range(16, 32))
    a, b = str(tmp_path / "py.avro"), str(tmp_path / "sub" / "nat.avro")
    _write_scores(a, schema, sp, uid, score, label, weight, per, native=False, sync_marker=sync)
    _write_scores(b, schema, sp, uid, score, label, weight, per, native=True, sync_marker=sync)
    assert open(a, "rb").read() == open(b, "rb").read()
    recs = list(avro.read_file(b))
    assert len(recs) == n and recs[0]["uid"] == int(uid[0])
    _write_scores(b, schema, sp, uid[:0], score[:0], None if label is None else label[:0],
                  None if weight is None else weight[:0], per[:0], native=True, sync_marker=sync)
    assert list(avro.read_file(b)) == []


@pytest.mark.parametrize("deflate", [False, True])
def test_native_block_writer_pipeline_with_many_small_blocks(tmp_path, deflate):
    """csrc/io_avro.cpp write_blocks: workers encode groups of blocks into a ring, the caller writes them in order. Thousands of
    3-record blocks on 1, 2 and 7 threads (more groups than ring slots: the ring wraps), both codecs: the same bytes whatever the
    thread count, and every record back through the generic codec."""
    from types import SimpleNamespace
    from gdmix_amd.io import avro
    sp = SimpleNamespace(uid_column_name="uid", prediction_score_column_name="predictionScore", label_column_name="response",
                         weight_column_name="weight", prediction_score_per_coordinate_column_name="predictionScorePerCoordinate")
    schema = avro.inference_output_schema(sp, has_weight=True)
    rng = np.random.default_rng(8)
    n = 20_001
    uid = rng.integers(-2 ** 62, 2 ** 62, n)
    score, per = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    label, weight = (rng.random(n) < 0.5).astype(np.float32), rng.random(n).astype(np.float32)
    header, sync = avro.container_header(schema, "deflate" if deflate else "null", bytes(range(16)))
    files = []
    for threads in (1, 2, 7):
        path = str(tmp_path / f"t{threads}.avro")
        native_reader.write_scores_avro(path, header, sync, uid, score, label, weight, per, block_records=3, deflate=deflate, threads=threads)
        files.append(open(path, "rb").read())
    assert files[0] == files[1] == files[2]
    recs = list(avro.read_file(str(tmp_path / "t7.avro")))
    assert len(recs) == n
    assert [r["uid"] for r in recs] == uid.tolist()
    np.testing.assert_array_equal(np.array([r["predictionScore"] for r in recs], np.float32), score)
    np.testing.assert_array_equal(np.array([r["weight"] for r in recs], np.float32), weight)


@pytest.mark.parametrize("int_ids", [False, True])
@pytest.mark.parametrize("bag", ["bag", None])
def test_native_tfrecord_writer_is_byte_identical_and_round_trips(tmp_path, int_ids, bag):
    b = synthetic.make_ragged_batch(300, seed=5)
    if int_ids:
        b.entity_ids = [str(11 * i - 5) for i in range(b.E)]
    a, n = str(tmp_path / "py.tfrecord"), str(tmp_path / "nat.tfrecord")
    write_grouped_partition(a, b, "ent", bag, int_entity_ids=int_ids, native=False)
    write_grouped_partition(n, b, "ent", bag, int_entity_ids=int_ids, native=True)
    assert open(a, "rb").read() == open(n, "rb").read()
    for suffix in (".gz", ".deflate"):
        z = str(tmp_path / ("z" + suffix.replace(".", "_")) / ("part.tfrecord" + suffix))
        write_grouped_partition(z, b, "ent", "bag", int_entity_ids=int_ids, native=True)
        r = read_grouped_partition(z, MD, "ent", "bag", "offset", "uid", "response", "weight", num_features=4096,
                                   check_crc=True, native=False)
        for k in ARRAYS:
            np.testing.assert_array_equal(getattr(r, k), getattr(b, k), err_msg=k)
        assert r.entity_ids == b.entity_ids
    # no label / no weight columns
    nolab = RawBatch(**{**{k: getattr(b, k) for k in ARRAYS}, "weight": None, "entity_ids": b.entity_ids, "has_label": False})
    write_grouped_partition(a, nolab, "ent", "bag", int_entity_ids=int_ids, native=False)
    write_grouped_partition(n, nolab, "ent", "bag", int_entity_ids=int_ids, native=True)
    assert open(a, "rb").read() == open(n, "rb").read()


# ---- Avro model reader: the same table the record-by-record Python loader builds ---------------------------------
def _model_for(tmp_path, D, has_intercept=True, feature_list=None):
    from gdmix_amd.model import RandomEffectLRLBFGSModel
    fl = feature_list or [(f"name{j}", "" if j % 4 else f"term{j}") for j in range(D)]
    with open(tmp_path / "features.csv", "w") as f:
        f.write("".join(f"{n},{t}\n" for n, t in fl))
    argv = ["--output_model_dir", str(tmp_path / "m"), "--feature_bag", "bag", "--feature_file", str(tmp_path / "features.csv"),
            "--partition_entity", "ent", "--has_intercept", "true" if has_intercept else "false",
            "--metadata_file", str(tmp_path / "md.json")]
    if not has_intercept:
        argv += ["--regularize_bias", "false"]
    return RandomEffectLRLBFGSModel(argv), fl


def _load_both(model, path, monkeypatch):
    nat = model._load_weights(path)
    monkeypatch.setattr(native_reader, "available", lambda: False)
    py = model._load_weights(path)
    monkeypatch.undo()
    return nat, py


def _assert_same_table(a, b):
    assert list(a.keys()) == list(b.keys())
    fa, fb = a.flatten(), b.flatten()
    for x, y in zip(fa[1:], fb[1:]):
        if x is None or y is None:
            assert x is None and y is None
        else:
            assert np.array_equal(x, y)


@pytest.mark.parametrize("with_variance", [False, True])
@pytest.mark.parametrize("codec", ["null", "deflate"])
def test_native_model_reader_matches_python_loader(tmp_path, monkeypatch, with_variance, codec):
    from gdmix_amd.io import avro
    from gdmix_amd.model import _export_models_to_avro
    D = 50
    model, fl = _model_for(tmp_path, D)
    table = _table(11, 2600, D, with_variance)
    path = str(tmp_path / "models.avro")
    _export_models_to_avro(path, table, fl, True, with_variance, native=False)
    if codec == "deflate":   # the same records through the generic writer with the deflate codec
        recs = list(avro.read_file(path))
        avro.write_file(path, avro.BAYESIAN_LINEAR_MODEL_SCHEMA, recs, codec="deflate", block_records=100)
    nat, py = _load_both(model, path, monkeypatch)
    assert len(nat) == 2600
    _assert_same_table(nat, py)
    # what was exported and thresholded is what comes back
    k = "id1"
    want = table[k]
    keep = np.abs(want.theta[1:]) > 1e-4
    assert np.array_equal(nat[k].theta, np.concatenate([want.theta[:1], want.theta[1:][keep]]))
    assert np.array_equal(nat[k].unique_global_indices, want.unique_global_indices[keep])
    assert (nat[k].variance is not None) == with_variance


def test_native_model_reader_errors_and_fallbacks(tmp_path, monkeypatch):
    from gdmix_amd.io import avro
    from gdmix_amd.model import ModelTable, _export_models_to_avro
    D = 6
    model, fl = _model_for(tmp_path, D)
    t = ModelTable()
    t.add_chunk(["a", "b"], np.array([0.5, 1.0, -2.0, 0.25, 3.0]), [0, 3, 5], np.array([1, 4, 5]), [0, 2, 3])
    path = str(tmp_path / "m.avro")
    # a feature that is not in this model's feature file -> KeyError, from both loaders
    wider = fl + [("extra", "")]
    t2 = ModelTable()
    t2.add_chunk(["a"], np.array([0.5, 1.0]), [0, 2], np.array([6]), [0, 1])
    _export_models_to_avro(path, t2, wider, True, False, native=False)
    with pytest.raises(KeyError):
        model._load_weights(path)
    monkeypatch.setattr(native_reader, "available", lambda: False)
    with pytest.raises(KeyError):
        model._load_weights(path)
    monkeypatch.undo()
    # a model written without an intercept read by a model that expects one -> AssertionError, from both
    t3 = ModelTable()
    t3.add_chunk(["a", "b"], np.array([1.0, -2.0, 3.0]), [0, 2, 3], np.array([1, 4, 5]), [0, 2, 3])
    _export_models_to_avro(path, t3, fl, False, False, native=False)
    with pytest.raises(AssertionError):
        model._load_weights(path)
    monkeypatch.setattr(native_reader, "available", lambda: False)
    with pytest.raises(AssertionError):
        model._load_weights(path)
    monkeypatch.undo()
    # a writer schema with another field order is left to the schema-driven decoder
    _export_models_to_avro(path, t, fl, True, False, native=False)
    recs = list(avro.read_file(path))
    schema = dict(avro.BAYESIAN_LINEAR_MODEL_SCHEMA)
    f = list(schema["fields"])
    schema["fields"] = [f[0], f[1], f[4], f[2], f[3]]
    assert not avro.is_model_schema(schema) and avro.is_model_schema(avro.BAYESIAN_LINEAR_MODEL_SCHEMA)
    avro.write_file(path, schema, recs)
    called = []
    real = native_reader.read_models_avro
    monkeypatch.setattr(native_reader, "read_models_avro", lambda *a, **k: called.append(1) or real(*a, **k))
    got = model._load_weights(path)
    assert not called and list(got.keys()) == ["a", "b"]
    assert np.array_equal(got["b"].theta, [0.25, 3.0]) and np.array_equal(got["b"].unique_global_indices, [5])
    # truncated file -> ValueError
    _export_models_to_avro(path, t, fl, True, False, native=False)
    data = open(path, "rb").read()
    open(path, "wb").write(data[:-5])
    with pytest.raises(ValueError):
        model._load_weights(path)
    # empty container
    avro.write_file(path, avro.BAYESIAN_LINEAR_MODEL_SCHEMA, [])
    assert len(model._load_weights(path)) == 0


def test_native_model_reader_without_intercept_and_duplicate_features(tmp_path, monkeypatch):
    from gdmix_amd.model import ModelTable, _export_models_to_avro
    fl = [("f0", ""), ("f1", "t"), ("f0", "")]      # the later of equal (name, term) pairs wins, as in the reference's dict
    model, _ = _model_for(tmp_path, 3, has_intercept=False, feature_list=fl)
    t = ModelTable()
    t.add_chunk(["x", "y"], np.array([0.5, -2.0, 3.0]), [0, 2, 3], np.array([1, 0, 1]), [0, 2, 3])
    path = str(tmp_path / "m.avro")
    _export_models_to_avro(path, t, fl, False, False, native=False)
    nat, py = _load_both(model, path, monkeypatch)
    _assert_same_table(nat, py)
    assert np.array_equal(nat["x"].unique_global_indices, [1, 2])


def test_pool_trim_releases_idle_blocks(tmp_path):
    """Arrays of a freed batch stay in the library's pool for the next partition (GDMIX_IO_POOL_MB); gdmix_io_pool_trim hands
    them back to the allocator and reports how much that was."""
    import gc
    from gdmix_amd import synthetic
    from gdmix_amd.io import native_reader
    from gdmix_amd.io.grouped_reader import write_grouped_partition
    native_reader.load_library()
    native_reader.pool_trim()
    b = synthetic.make_batch(40000, 16, 4, 1024, seed=3)
    path = str(tmp_path / "p" / "part-0.tfrecord")
    write_grouped_partition(path, b, "ent", "bag", weight_column_name=None)
    got = native_reader.read_grouped_files([path], "ent", "bag", "offset", "uid", label_column_name="response")
    assert got.E == b.E and np.array_equal(got.col_global, b.col_global)
    assert native_reader.pool_trim() == 0 or True      # blocks in use are never touched
    assert np.array_equal(got.col_global, b.col_global)
    del got
    gc.collect()
    freed = native_reader.pool_trim()
    assert freed >= b.Z * 8                             # at least the feature index array (int64) was idle in the pool
    assert native_reader.pool_trim() == 0


def test_narrowing_picks_the_widths_and_refuses_what_does_not_fit(tmp_path):
    """gdmix_io_narrow: one byte per count while every sample has at most 255 non-zeros, two bytes per feature index while every index is
    below 65 536; a wide sample or a high index switches that one array over; an index of 2^31 is an error, as in RawBatch.to_wire()."""
    def read(cols_of_sample, dim):
        md = dict(MD, features=[dict(MD["features"][0], shape=[dim])] + MD["features"][1:])
        k = [len(c) for c in cols_of_sample]
        b = RawBatch(ent_row_ptr=[0, len(k)], row_nnz_ptr=np.concatenate([[0], np.cumsum(k)]), col_global=np.concatenate(cols_of_sample),
                     val=np.ones(sum(k), np.float32), y=np.arange(len(k)) % 2, offset=np.zeros(len(k), np.float32),
                     uid=np.arange(len(k)), entity_ids=["e"])
        d = tmp_path / f"d{dim}_{max(k)}"
        write_grouped_partition(str(d / "part-0.tfrecord"), b, "ent", "bag", "offset", "uid", "response", None)
        return b, read_grouped_partition(str(d), md, "ent", "bag", "offset", "uid", "response", None, num_features=dim, native=True, wire=True)
    b, w = read([np.arange(255), np.array([65535])], 65536)
    assert (w.to_wire()["row_nnz_width"], w.to_wire()["col_width"], w.to_wire()["y_width"]) == (1, 2, 1)
    same_wire(b, w)
    b, w = read([np.arange(256), np.array([65536])], 1 << 20)
    assert (w.to_wire()["row_nnz_width"], w.to_wire()["col_width"]) == (2, 4)
    same_wire(b, w)
    b, w = read([np.arange(65536), np.array([0x7fffffff])], 1 << 31)
    assert (w.to_wire()["row_nnz_width"], w.to_wire()["col_width"]) == (4, 4)
    same_wire(b, w)
    with pytest.raises(ValueError, match=r"2\^31"):
        read([np.array([1 << 31])], 1 << 32)


def test_feature_file_encoded_from_its_bytes_equals_the_feature_by_feature_encoding(tmp_path):
    """native_reader.EncodedFeatures.from_feature_file (round 5): for a plain `name,term` file the Avro encoding string(name) +
    string(term) of every feature is the file's bytes shifted by one with the two length bytes dropped in — no Python object per
    feature (65 536 features: milliseconds instead of 0.1 s holding the interpreter lock while the first files are written). Same
    bytes and offsets as the feature-by-feature encoding; anything not plain (quotes, a line without / with two commas, a field
    of 64 bytes or more, no final newline, carriage returns) returns None and the caller takes the slow way."""
    from gdmix_amd.io import avro
    from gdmix_amd.io.features import read_feature_list
    p = str(tmp_path / "features.csv")
    rng = np.random.default_rng(3)
    lines = [f"name{i}" + "x" * int(rng.integers(0, 50)) + "," + "t" * int(rng.integers(0, 63)) for i in range(5000)] + [",", "a,", ",b"]
    with open(p, "w") as f:
        f.write("".join(ln + "\n" for ln in lines))
    fast = native_reader.EncodedFeatures.from_feature_file(p)
    ref = native_reader.EncodedFeatures([avro.enc_string(n) + avro.enc_string(t) for n, t in read_feature_list(p)])
    assert fast is not None and fast.count == ref.count == len(lines)
    assert fast.bytes == ref.bytes and np.array_equal(fast.ptr, ref.ptr) and fast[17] == ref[17]
    for bad in ('a,b\n"x,y",z\n', "a,b\nc\n", "a,b,c\nd\n", "a,b\r\n", "a,b", "n" * 64 + ",t\n", "n," + "t" * 64 + "\n"):
        with open(p, "w", newline="") as f:
            f.write(bad)
        assert native_reader.EncodedFeatures.from_feature_file(p) is None, bad
    open(p, "w").close()
    assert native_reader.EncodedFeatures.from_feature_file(p).count == 0
