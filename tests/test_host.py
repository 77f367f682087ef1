"""CPU tests of the host-side mirror of the reference's RE interface: argv/params, TFRecord and Avro codecs,
metadata, driver partition striding / file naming, model train -> Avro -> predict (through a test double of
the device solver built on the oracle), warm start, prior carry-over."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, OracleSolverDouble, load_fixture
from gdmix_amd import constants
from gdmix_amd.batch import RawBatch
from gdmix_amd.driver import RandomEffectDriver
from gdmix_amd.io import avro, tfrecord
from gdmix_amd.io.grouped_reader import read_grouped_partition, resolve_input_files, write_grouped_partition
from gdmix_amd.io.metadata import DatasetMetadata
from gdmix_amd.model import ModelTable, RandomEffectLRLBFGSModel, TrainingResult, _model_coefficients_for_batch
from gdmix_amd.params import Params, REParams, SchemaParams

RES = os.path.join(GOLDEN, "ref_resources")


# ---- B1: CLI ------------------------------------------------------------------------------------------------
# the job dicts gdmix-workflow hands to `python -m gdmix.gdmix` (gdmix-workflow/test/test_workflow_generator.py:189-222)
WORKFLOW_PARAMS = {'uid_column_name': 'uid', 'weight_column_name': 'weight', 'label_column_name': 'response',
                   'prediction_score_column_name': 'predictionScore',
                   'prediction_score_per_coordinate_column_name': 'predictionScorePerCoordinate', 'action': 'train',
                   'stage': 'random_effect', 'model_type': 'logistic_regression',
                   'training_score_dir': 'lr-training/per-user/training_scores',
                   'validation_score_dir': 'lr-training/per-user/validation_scores',
                   'partition_list_file': 'lr-training/per-user/partition/partitionList.txt', '__frozen__': True}
WORKFLOW_MODEL = {'metadata_file': 'lr-training/per-user/partition/metadata/tensor_metadata.json',
                  'output_model_dir': 'lr-training/per-user/models',
                  'training_data_dir': 'lr-training/per-user/partition/trainingData',
                  'validation_data_dir': 'lr-training/per-user/partition/validationData', 'feature_bag': 'per_user',
                  'feature_file': 'movieLens/per_user/featureList/per_user', 'regularize_bias': False, 'l2_reg_weight': 1.0,
                  'lbfgs_tolerance': 1e-12, 'num_of_lbfgs_curvature_pairs': 10, 'num_of_lbfgs_iterations': 100,
                  'has_intercept': True, 'offset_column_name': 'offset', 'batch_size': 16, 'data_format': 'tfrecord',
                  'partition_entity': 'user_id', 'enable_local_indexing': False, 'max_training_queue_size': 10,
                  'training_queue_timeout_in_seconds': 300, 'num_of_consumers': 1, 'random_effect_variance_mode': None,
                  'disable_random_effect_scoring_after_training': False, '__frozen__': True}


def _local_ops_argv(*dicts):
    """single_node/local_ops.py:15-23: every non-None item becomes --k=v."""
    argv = ["gdmix"]
    for d in dicts:
        argv += [f"--{k}={v}" for k, v in d.items() if v is not None and k != "__frozen__"]
    return argv


def test_workflow_argv_parses_into_all_three_param_classes():
    argv = _local_ops_argv(WORKFLOW_PARAMS, WORKFLOW_MODEL)
    p, s, m = Params.__from_argv__(argv), SchemaParams.__from_argv__(argv), REParams.__from_argv__(argv)
    for k, v in WORKFLOW_PARAMS.items():
        if k != "__frozen__":
            assert getattr(p, k) == v
    for k, v in WORKFLOW_MODEL.items():
        if k != "__frozen__":
            assert getattr(m, k) == v, k
    assert s.uid_column_name == "uid" and s.prediction_score_per_coordinate_column_name == "predictionScorePerCoordinate"


def test_argv_forms_and_assertions():
    a = ["--metadata_file", "m", "--output_model_dir=o", "--has_intercept", "False", "--regularize_bias=False",
         "--feature_bag", "b", "--unknown", "1"]
    m = REParams.__from_argv__(a)
    assert m.has_intercept is False and m.regularize_bias is False and m.l2_reg_weight == 1.0 and m.batch_size == 16
    # REParams.__post_init__ does not chain to LRParams' checks upstream (random_effect_lr_lbfgs_model.py:48-53):
    # no intercept with regularize_bias left at its default True is accepted by the random-effect stage ...
    m = REParams.__from_argv__(["--metadata_file", "m", "--output_model_dir", "o", "--has_intercept", "False", "--feature_bag", "b"])
    assert m.has_intercept is False and m.regularize_bias is True
    from gdmix_amd.fe_model import FixedLRParams
    with pytest.raises(AssertionError):   # ... and rejected by the fixed-effect stage: "Intercept must be used when it is regularized"
        FixedLRParams.__from_argv__(["--metadata_file", "m", "--output_model_dir", "o", "--has_intercept", "False", "--feature_bag", "b"])
    with pytest.raises(AssertionError):   # queue size must exceed consumers
        REParams.__from_argv__(["--metadata_file", "m", "--output_model_dir", "o", "--num_of_consumers", "10"])
    with pytest.raises(AssertionError):
        REParams.__from_argv__(["--metadata_file", "m", "--output_model_dir", "o", "--random_effect_variance_mode", "bogus"])
    with pytest.raises(ValueError):
        REParams.__from_argv__(["--output_model_dir", "o"])
    with pytest.raises(AssertionError):   # train needs a label column
        Params.__from_argv__(["--uid_column_name", "uid", "--action", "train"])
    assert Params.__from_argv__(["--uid_column_name", "u", "--action", "inference", "--prediction_score_column_name", "s"]).action == "inference"


# ---- TFRecord ----------------------------------------------------------------------------------------------
def test_reference_fixture_decodes_to_the_documented_tensors():
    """test/resources/grouped_per_member_train/data.tfrecord vs the layout in job_consumers.py:176-199."""
    b = read_grouped_partition(os.path.join(RES, "data.tfrecord"), os.path.join(RES, "data.json"), "memberId", "per_member",
                               "offset", "uid", "response", "weight", num_features=100, check_crc=True)
    assert b.entity_ids == ["100034", "100"]
    assert b.ent_row_ptr.tolist() == [0, 2, 3] and b.row_nnz_ptr.tolist() == [0, 5, 7, 9]
    assert b.col_global.tolist() == [0, 7, 60, 80, 95, 34, 57, 10, 11]
    np.testing.assert_array_equal(b.val, np.array([1, 2, 3, 5, 6.6, 1, 2, -3.5, 2.3], np.float32))
    assert b.y.tolist() == [0, 1, 1] and b.uid.tolist() == [10, 20, 23] and b.weight.tolist() == [1, 2, 1]
    np.testing.assert_array_equal(b.offset, np.array([0.5, 0.75, 0.2], np.float32))


@pytest.mark.parametrize("suffix", [".tfrecord", ".tfrecord.gz", ".tfrecord.deflate"])
def test_tfrecord_round_trip_all_compressions(tmp_path, suffix):
    b, _, _, _ = load_fixture("ragged")
    path = str(tmp_path / ("part-0" + suffix))
    write_grouped_partition(path, b, "ent", "bag")
    md = {"features": [{"name": "bag", "dtype": "float", "shape": [200], "isSparse": True},
                       {"name": "weight", "dtype": "float", "shape": [], "isSparse": False},
                       {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                       {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                       {"name": "ent", "dtype": "string", "shape": [], "isSparse": False}],
          "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}
    assert resolve_input_files(str(tmp_path)) == [path]
    r = read_grouped_partition(str(tmp_path), md, "ent", "bag", "offset", "uid", "response", "weight", num_features=200,
                               check_crc=True)
    for k in ("ent_row_ptr", "row_nnz_ptr", "col_global", "val", "y", "offset", "weight", "uid"):
        np.testing.assert_array_equal(getattr(r, k), getattr(b, k))
    assert r.entity_ids == b.entity_ids


def test_crc_and_masking_known_answers():
    assert tfrecord.crc32c(b"123456789") == 0xE3069283          # CRC-32C check value
    assert tfrecord.crc32c(b"") == 0
    with pytest.raises(ValueError):
        data = bytearray(open(os.path.join(RES, "data.tfrecord"), "rb").read())
        data[40] ^= 1
        p = os.path.join(os.path.dirname(__file__), "_corrupt.tfrecord")
        try:
            open(p, "wb").write(bytes(data))
            list(tfrecord.iter_records(p, check_crc=True))
        finally:
            os.remove(p)


def test_unpacked_protobuf_lists_are_accepted():
    # a writer may emit repeated scalars unpacked: int64 value = 1 (wire type 0), float (wire type 5)
    feat_i = b"\x1a" + bytes([4]) + b"\x08\x05\x08\x07"                     # Int64List{value:5, value:7} unpacked
    feat_f = b"\x12" + bytes([5]) + b"\x0d" + np.float32(1.5).tobytes()      # FloatList{value:1.5} unpacked
    assert tfrecord._decode_feature(memoryview(feat_i))[1].tolist() == [5, 7]
    assert tfrecord._decode_feature(memoryview(feat_f))[1].tolist() == [1.5]


# ---- metadata -------------------------------------------------------------------------------------------------
def test_metadata_valid_and_invalid():
    md = DatasetMetadata(os.path.join(RES, "data.json"))
    assert md.get_feature_shape("per_member") == [100]
    assert md.get_feature_names() == ["per_member", "weight", "offset", "uid", "memberId"] and md.get_label_names() == ["response"]
    with pytest.raises(ValueError):
        DatasetMetadata({"features": [{"name": "a", "dtype": "int", "shape": [], "isSparse": False},
                                      {"name": "a", "dtype": "int", "shape": [], "isSparse": False}]})
    with pytest.raises(ValueError):
        DatasetMetadata({"features": [{"name": "a", "dtype": "complex", "shape": [], "isSparse": False}]})
    with pytest.raises(ValueError):
        DatasetMetadata({"features": [{"name": "a", "dtype": "int", "shape": None, "isSparse": False}]})
    with pytest.raises(TypeError):
        DatasetMetadata({"features": {"name": "a"}})


# ---- Avro ----------------------------------------------------------------------------------------------------
def test_model_export_matches_gen_one_avro_model_known_answers(tmp_path):
    """Literal records of test/util/test_io_utils.py:86-188 (threshold and variance variants)."""
    feature_list = [("f1,2", "t1"), ("f2", ""), ("f3", "t3,3")]
    cls = constants.PHOTON_LR_MODEL_CLASS
    t = ModelTable()
    t.update({"1234": TrainingResult(np.array([7.8, 1.2, 3.4, 5.6]), None, np.arange(3))})
    from gdmix_amd.model import _export_models_to_avro
    p = str(tmp_path / "m.avro")
    _export_models_to_avro(p, t, feature_list, True, False, sparsity_threshold=0.0)
    assert list(avro.read_file(p)) == [{"modelId": "1234", "modelClass": cls, "means": [
        {"name": "(INTERCEPT)", "term": "", "value": 7.8}, {"name": "f1,2", "term": "t1", "value": 1.2},
        {"name": "f2", "term": "", "value": 3.4}, {"name": "f3", "term": "t3,3", "value": 5.6}], "variances": None, "lossFunction": ""}]
    t = ModelTable()
    t.update({"1234": TrainingResult(np.array([0.8, 1.2, 3.4, -5.6]), None, np.arange(3))})
    _export_models_to_avro(p, t, feature_list, True, False, sparsity_threshold=3.4)
    assert list(avro.read_file(p))[0]["means"] == [{"name": "(INTERCEPT)", "term": "", "value": 0.8},
                                                   {"name": "f3", "term": "t3,3", "value": -5.6}]
    t = ModelTable()
    t.update({"1234": TrainingResult(np.array([-7.8, 1.2, 3.4, 5.6]), np.array([1.2, 7.8, 9.0, 10.1]), np.arange(3))})
    _export_models_to_avro(p, t, feature_list, True, True, sparsity_threshold=0.0)
    rec = list(avro.read_file(p))[0]
    assert [v["value"] for v in rec["variances"]] == [1.2, 7.8, 9.0, 10.1] and rec["variances"][1]["name"] == "f1,2"


def test_avro_generic_codec_round_trip_and_deflate(tmp_path):
    schema = {"type": "record", "name": "r", "fields": [{"name": "a", "type": "long"}, {"name": "b", "type": ["null", "float"], "default": None},
                                                        {"name": "c", "type": {"type": "map", "values": "string"}}]}
    recs = [{"a": -(2 ** 40), "b": None, "c": {"k": "v"}}, {"a": 3, "b": 0.5, "c": {}}]
    for codec in ("null", "deflate"):
        p = str(tmp_path / f"x_{codec}.avro")
        assert avro.write_file(p, schema, recs, codec=codec, block_records=1) == 2
        assert list(avro.read_file(p)) == recs


# ---- driver ---------------------------------------------------------------------------------------------------
class _MockModel:
    def __init__(self, base):
        self.checkpoint_path = os.path.join(base, "model")
        self.training_data_dir = os.path.join(base, "train", "active")
        self.passive_training_data_dir = os.path.join(base, "train", "passive")
        self.validation_data_dir = os.path.join(base, "valid")
        self.metadata_file = "meta.json"
        self.calls = []

    def train(self, **kw):
        self.calls.append(("train", kw))

    def predict(self, **kw):
        self.calls.append(("predict", kw))

    def export(self, **kw):
        self.calls.append(("export", kw))


def _base_params(tmp_path, action="train"):
    return Params(uid_column_name="uid", weight_column_name="weight", label_column_name="response",
                  prediction_score_column_name="predictionScore", action=action, stage="random_effect",
                  training_score_dir=str(tmp_path / "ts"), validation_score_dir=str(tmp_path / "vs"),
                  partition_list_file=os.path.join(RES, "partition_list.txt"))


def test_driver_without_tf_config_is_single_local_worker(tmp_path, monkeypatch):
    monkeypatch.delenv("TF_CONFIG", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    d = RandomEffectDriver(_base_params(tmp_path), _MockModel(str(tmp_path)))
    assert d.execution_context == {"task_type": "worker", "task_index": 0, "cluster_spec": None, "num_workers": 1,
                                   "num_shards": 1, "shard_index": 0, "is_chief": True}


@pytest.mark.parametrize("worker_index", [0, 1, 4])
def test_driver_partition_striding_and_file_names_with_tf_config(tmp_path, monkeypatch, worker_index):
    tf_config = {"task": {"type": "worker", "index": worker_index},
                 "cluster": {"worker": [f"node{i}.example.com:1" for i in range(5)], "evaluator": ["node6.example.com:2"]}}
    monkeypatch.setenv("TF_CONFIG", json.dumps(tf_config))
    model = _MockModel(str(tmp_path))
    d = RandomEffectDriver(_base_params(tmp_path), model)
    assert "TF_CONFIG" not in os.environ                      # random effect runs in local mode
    ctx = d.execution_context
    assert (ctx["task_index"], ctx["num_workers"], ctx["num_shards"], ctx["shard_index"]) == (worker_index, 5, 1, 0)
    allp = [int(x) for x in open(os.path.join(RES, "partition_list.txt")).readline().split(",")]
    mine = allp[worker_index::5]
    assert d._get_partition_list() == mine
    for k in mine:
        os.makedirs(os.path.join(model.training_data_dir, f"partitionId={k}", "x"))
    d.run_training(SchemaParams(uid_column_name="uid"), export_model=False)
    assert [c[1]["training_data_dir"] for c in model.calls] == [os.path.join(model.training_data_dir, f"partitionId={k}") for k in mine]
    kw = model.calls[0][1]
    k0 = mine[0]
    assert kw["checkpoint_path"] == os.path.join(model.checkpoint_path, f"partitionId={k0}")
    assert kw["validation_data_dir"] == os.path.join(model.validation_data_dir, f"partitionId={k0}")
    ec = kw["execution_context"]
    assert ec["partition_index"] == k0
    assert ec["active_training_output_file"] == str(tmp_path / "ts" / f"partitionId={k0}" / f"part-{worker_index:05d}-active.avro")
    assert ec["passive_training_output_file"] == str(tmp_path / "ts" / f"partitionId={k0}" / f"part-{worker_index:05d}-passive.avro")
    assert ec["validation_output_file"] == str(tmp_path / "vs" / f"partitionId={k0}" / f"part-{worker_index:05d}.avro")
    assert "passive_training_data_dir" not in ec                # no passive data on disk


def test_driver_skips_empty_partition_directories(tmp_path, monkeypatch):
    monkeypatch.delenv("TF_CONFIG", raising=False)
    model = _MockModel(str(tmp_path))
    d = RandomEffectDriver(_base_params(tmp_path), model)
    for k in d._get_partition_list():
        os.makedirs(os.path.join(model.training_data_dir, f"partitionId={k}"))
    d.run_training(SchemaParams(uid_column_name="uid"))
    assert model.calls == []


# ---- model: train -> avro -> predict through the oracle double ----------------------------------------------------
def _raw_params(tmp_path, extra=()):
    return ["--uid_column_name", "uid", "--weight_column_name", "weight", "--label_column_name", "response",
            "--training_data_dir", str(tmp_path / "data"), "--validation_data_dir", str(tmp_path / "valid"),
            "--output_model_dir", str(tmp_path / "models"), "--metadata_file", os.path.join(RES, "data.json"),
            "--feature_bag", "per_member", "--feature_file", os.path.join(RES, "fake_feature_file.csv"),
            "--partition_entity", "memberId", "--l2_reg_weight", "0.1", "--has_intercept", "True"] + list(extra)


def _make_model(tmp_path, extra=()):
    m = RandomEffectLRLBFGSModel(_raw_params(tmp_path, extra))
    m._solver = OracleSolverDouble()
    return m


def _fixture_dir(tmp_path, name="train"):
    d = tmp_path / name / "partitionId=0"
    os.makedirs(d)
    import shutil
    shutil.copy(os.path.join(RES, "data.tfrecord"), d / "data.tfrecord")
    return str(d)


def _train_ctx(tmp_path, with_outputs=True):
    ctx = {"partition_index": 0, "task_index": 0, "num_workers": 1, "is_chief": True}
    if with_outputs:
        ctx.update(validation_output_file=str(tmp_path / "vs" / "part-00000.avro"),
                   active_training_output_file=str(tmp_path / "ts" / "part-00000-active.avro"),
                   passive_training_output_file=str(tmp_path / "ts" / "part-00000-passive.avro"))
    return ctx


SCHEMA = SchemaParams(uid_column_name="uid", weight_column_name="weight", label_column_name="response",
                      prediction_score_column_name="predictionScore")


def test_model_attributes_follow_the_reference_constructor(tmp_path):
    m = _make_model(tmp_path)
    assert m.training_data_dir == os.path.join(str(tmp_path / "data"), "active")
    assert m.passive_training_data_dir == os.path.join(str(tmp_path / "data"), "passive")
    assert m.validation_data_dir == str(tmp_path / "valid") and m.checkpoint_path == str(tmp_path / "models")
    assert m.metadata_file == os.path.join(RES, "data.json") and m.export("x") is None


def test_train_writes_model_and_scores_equal_to_cold_prediction(tmp_path):
    """Reference integration test (test_random_effect_lr_lbfgs_model.py:82-152): scoring while training must
    equal a cold predict() with the saved model, record for record; coefficients equal the golden fixture."""
    m = _make_model(tmp_path)
    train_dir = _fixture_dir(tmp_path)
    m.train(train_dir, train_dir, m.metadata_file, str(tmp_path / "models"), _train_ctx(tmp_path), SCHEMA)
    model_file = str(tmp_path / "models" / "part-00000.avro")
    recs = list(avro.read_file(model_file))
    assert [r["modelId"] for r in recs] == ["100034", "100"]
    _, _, exp, _ = load_fixture("ref_fixture_l2_0.1")
    got0 = [c["value"] for c in recs[0]["means"]]
    np.testing.assert_allclose(got0, exp["theta_thr"][:8], rtol=1e-9)
    assert recs[0]["means"][0]["name"] == "(INTERCEPT)" and recs[0]["means"][1] == {"name": "f1", "term": "t1", "value": got0[1]}
    assert recs[0]["means"][2]["name"] == "f8" and recs[0]["lossFunction"] == "" and recs[0]["variances"] is None
    active = list(avro.read_file(str(tmp_path / "ts" / "part-00000-active.avro")))
    valid = list(avro.read_file(str(tmp_path / "vs" / "part-00000.avro")))
    assert [r["uid"] for r in active] == [10, 20, 23] and active == valid
    assert set(active[0]) == {"uid", "predictionScore", "response", "weight", "predictionScorePerCoordinate"}
    assert active[0]["response"] == 0.0 and active[1]["weight"] == 2.0
    out = tmp_path / "cold"
    m2 = _make_model(tmp_path)
    m2.predict(str(out), train_dir, m2.metadata_file, str(tmp_path / "models"), {"partition_index": 0}, SCHEMA)
    assert list(avro.read_file(str(out / "part-00000.avro"))) == active
    assert not os.path.exists(str(tmp_path / "ts" / "part-00000-passive.avro"))   # no passive dir in the context


def test_predict_requires_a_model_file(tmp_path):
    m = _make_model(tmp_path)
    with pytest.raises(FileNotFoundError):
        m.predict(str(tmp_path / "o"), _fixture_dir(tmp_path), m.metadata_file, str(tmp_path / "nomodels"),
                  {"partition_index": 0}, SCHEMA)


def test_bad_entity_column_fails_loudly(tmp_path):
    m = _make_model(tmp_path, ["--partition_entity", "bogus"])
    d = _fixture_dir(tmp_path)
    with pytest.raises(ValueError):
        m.train(d, None, m.metadata_file, str(tmp_path / "models"), _train_ctx(tmp_path, False), SCHEMA)


@pytest.mark.parametrize("local", ["True", "False"])
def test_warm_start_is_a_fixed_point_and_cold_one_iteration_differs(tmp_path, local):
    """test_random_effect_lr_lbfgs_model.py:231-350: train; reload + 1 more L-BFGS iteration stays put;
    a cold 1-iteration run lands somewhere else."""
    d = _fixture_dir(tmp_path)
    m = _make_model(tmp_path, ["--enable_local_indexing", local])
    m.train(d, None, m.metadata_file, str(tmp_path / "models"), _train_ctx(tmp_path, False), SCHEMA)
    first = {r["modelId"]: r for r in avro.read_file(str(tmp_path / "models" / "part-00000.avro"))}
    w = _make_model(tmp_path, ["--num_of_lbfgs_iterations", "1", "--enable_local_indexing", local])
    w.train(d, None, w.metadata_file, str(tmp_path / "models"), _train_ctx(tmp_path, False), SCHEMA)
    warm = {r["modelId"]: r for r in avro.read_file(str(tmp_path / "models" / "part-00000.avro"))}
    for k in first:
        np.testing.assert_allclose([c["value"] for c in warm[k]["means"]], [c["value"] for c in first[k]["means"]], rtol=1e-5, atol=1e-6)
    os.remove(str(tmp_path / "models" / "part-00000.avro"))
    c = _make_model(tmp_path, ["--num_of_lbfgs_iterations", "1"])
    c.train(d, None, c.metadata_file, str(tmp_path / "models"), _train_ctx(tmp_path, False), SCHEMA)
    cold = {r["modelId"]: r for r in avro.read_file(str(tmp_path / "models" / "part-00000.avro"))}
    assert not np.allclose([x["value"] for x in cold["100034"]["means"]], [x["value"] for x in first["100034"]["means"]], rtol=1e-3)


def test_prior_entities_missing_from_new_data_are_carried_over(tmp_path):
    d = _fixture_dir(tmp_path)
    prior = ModelTable()
    prior.update({"999": TrainingResult(np.array([0.5, -0.25]), None, np.array([3]))})
    from gdmix_amd.io.features import read_feature_list
    from gdmix_amd.model import _export_models_to_avro
    os.makedirs(tmp_path / "models")
    _export_models_to_avro(str(tmp_path / "models" / "part-00000.avro"), prior, read_feature_list(os.path.join(RES, "fake_feature_file.csv")), True, False)
    m = _make_model(tmp_path)
    m.train(d, None, m.metadata_file, str(tmp_path / "models"), _train_ctx(tmp_path, False), SCHEMA)
    recs = list(avro.read_file(str(tmp_path / "models" / "part-00000.avro")))
    assert [r["modelId"] for r in recs] == ["999", "100034", "100"]          # dict.update order
    assert recs[0]["means"] == [{"name": "(INTERCEPT)", "term": "", "value": 0.5}, {"name": "f4", "term": "t4", "value": -0.25}]


def test_intercept_only_model(tmp_path):
    """feature_bag absent: one dummy zero feature, theta = [b, 0], index [0] (test :138-152)."""
    params = [p for p in _raw_params(tmp_path)]
    i = params.index("--feature_bag")
    del params[i:i + 2]
    params[params.index("--metadata_file") + 1] = os.path.join(RES, "data_intercept_only.json")
    m = RandomEffectLRLBFGSModel(params)
    m._solver = OracleSolverDouble()
    assert m.feature_file is None
    d = _fixture_dir(tmp_path)
    m.train(d, None, m.metadata_file, str(tmp_path / "models"), _train_ctx(tmp_path, False), SCHEMA)
    recs = list(avro.read_file(str(tmp_path / "models" / "part-00000.avro")))
    assert all(len(r["means"]) == 1 and r["means"][0]["name"] == "(INTERCEPT)" for r in recs)
    t = m._load_weights(str(tmp_path / "models" / "part-00000.avro"))
    for k in t.keys():
        tr = t[k]
        assert len(tr.theta) == 2 and tr.theta[1] == 0.0 and tr.unique_global_indices.tolist() == [0]


def test_warm_start_mapping_matches_naive_dictionary_logic():
    """_model_coefficients_for_batch vs a literal transcription of prepare_jobs:262-288 semantics."""
    b, opts, exp, extra = load_fixture("warm_stage2")
    from oracle import oracle
    pk = oracle.pack(b.ent_row_ptr, b.row_nnz_ptr, b.col_global)
    table = ModelTable()
    pptr = extra["prior_feat_ptr"]
    E = b.E
    coef_ptr = pptr + np.arange(E + 1)
    table.add_chunk(b.entity_ids, extra["prior_theta"], coef_ptr, extra["prior_unique_global"], pptr)
    for native in (False, True):
        theta0, has_model = _model_coefficients_for_batch(table, b.entity_ids, pk["unique_global"], pk["ent_feat_ptr"], True, 1024,
                                                          native=native)
        assert has_model.all()
        np.testing.assert_array_equal(theta0, exp["theta0"])          # what the reference fed fmin_l_bfgs_b


@pytest.mark.parametrize("has_intercept", [True, False])
def test_native_coefficient_mapping_equals_the_numpy_one(has_intercept):
    """Several table chunks, entities without a model, models listing their features in any order (and twice)."""
    rng = np.random.default_rng(4)
    E, D = 500, 40
    ic = 1 if has_intercept else 0
    d = rng.integers(0, 12, E)
    feat_ptr = np.concatenate([[0], np.cumsum(d)]).astype(np.int64)
    uniq = np.concatenate([np.sort(rng.choice(D, k, replace=False)) for k in d] + [np.zeros(0, np.int64)]).astype(np.int64)
    ids = [f"e{i}" for i in range(E)]
    table = ModelTable()
    for c in range(3):
        own = [i for i in range(E) if i % 4 == c]              # i % 4 == 3: no model
        pd = rng.integers(0, 10, len(own))
        pf = np.concatenate([[0], np.cumsum(pd)]).astype(np.int64)
        pidx = rng.integers(0, D, pf[-1]) if c == 2 else np.concatenate(
            [np.sort(rng.choice(D, k, replace=False)) for k in pd] + [np.zeros(0, np.int64)]).astype(np.int64)
        table.add_chunk([ids[i] for i in own], rng.standard_normal(pf[-1] + len(own) * ic), pf + np.arange(len(own) + 1) * ic, pidx, pf)
    a, ha = _model_coefficients_for_batch(table, ids, uniq, feat_ptr, has_intercept, D, native=False)
    b, hb = _model_coefficients_for_batch(table, ids, uniq, feat_ptr, has_intercept, D, native=True)
    assert np.array_equal(ha, hb) and ha.sum() == sum(1 for i in range(E) if i % 4 != 3)
    np.testing.assert_array_equal(a, b)
    assert np.count_nonzero(a) > 100


# ---- host pipeline: read ahead, write behind ------------------------------------------------------------------------
def _pipeline_job(tmp_path, parts=(0, 1, 2)):
    import shutil
    from gdmix_amd.params import Params
    plist = tmp_path / "plist.txt"
    plist.write_text(",".join(str(k) for k in parts))
    for k in parts:
        for sub in ("data/active", "valid"):
            d = tmp_path / sub / f"partitionId={k}"
            os.makedirs(d)
            shutil.copy(os.path.join(RES, "data.tfrecord"), d / "data.tfrecord")
    base = Params(uid_column_name="uid", weight_column_name="weight", label_column_name="response",
                  prediction_score_column_name="predictionScore", action="train", stage="random_effect",
                  training_score_dir=str(tmp_path / "ts"), validation_score_dir=str(tmp_path / "vs"),
                  partition_list_file=str(plist))
    return base, _make_model(tmp_path)


def test_driver_pipeline_reads_each_partition_once_and_writes_the_same_files(tmp_path, monkeypatch):
    monkeypatch.delenv("TF_CONFIG", raising=False)
    base, model = _pipeline_job(tmp_path)
    reads = []
    real = model._read_files
    monkeypatch.setattr(model, "_read_files", lambda path, *a: reads.append(path) or real(path, *a))
    RandomEffectDriver(base, model).run_training(SCHEMA)
    assert model._io_pool is None and not model._pending_writes and not model._prefetched      # pipeline closed, nothing left
    # training data decoded once per partition (the scoring pass reuses it), validation data once
    assert sorted(reads) == sorted([str(tmp_path / "data" / "active" / f"partitionId={k}") for k in range(3)] +
                                   [str(tmp_path / "valid" / f"partitionId={k}") for k in range(3)])
    # the same job one partition at a time through the synchronous API
    ref = _make_model(tmp_path)
    ref.checkpoint_path = str(tmp_path / "models_ref")
    ref.model_params = ref.model_params.__class__(**{**ref.model_params.__dict__, "output_model_dir": str(tmp_path / "models_ref")})
    for k in range(3):
        ctx = {"partition_index": k, "validation_output_file": str(tmp_path / "vs_ref" / f"{k}.avro"),
               "active_training_output_file": str(tmp_path / "ts_ref" / f"{k}.avro")}
        ref.train(str(tmp_path / "data" / "active" / f"partitionId={k}"), str(tmp_path / "valid" / f"partitionId={k}"),
                  ref.metadata_file, None, ctx, SCHEMA)
        got = list(avro.read_file(str(tmp_path / "models" / f"part-{k:05d}.avro")))
        assert got == list(avro.read_file(str(tmp_path / "models_ref" / f"part-{k:05d}.avro"))) and len(got) > 0
        assert list(avro.read_file(str(tmp_path / "ts" / f"partitionId={k}" / "part-00000-active.avro"))) == \
            list(avro.read_file(str(tmp_path / "ts_ref" / f"{k}.avro")))
        assert list(avro.read_file(str(tmp_path / "vs" / f"partitionId={k}" / "part-00000.avro"))) == \
            list(avro.read_file(str(tmp_path / "vs_ref" / f"{k}.avro")))


def test_driver_pipeline_warm_start_loads_the_next_prior_model_ahead(tmp_path, monkeypatch):
    monkeypatch.delenv("TF_CONFIG", raising=False)
    base, model = _pipeline_job(tmp_path)
    RandomEffectDriver(base, model).run_training(SCHEMA)
    first = {k: list(avro.read_file(str(tmp_path / "models" / f"part-{k:05d}.avro"))) for k in range(3)}
    ahead = []
    real = model.prefetch_prior_model
    monkeypatch.setattr(model, "prefetch_prior_model", lambda k: ahead.append(k) or real(k))
    loads = []
    real_load = model._load_weights_from
    monkeypatch.setattr(model, "_load_weights_from", lambda f: loads.append(os.path.basename(f)) or real_load(f))
    RandomEffectDriver(base, model).run_training(SCHEMA)
    assert ahead == [1, 2] and sorted(loads) == [f"part-{k:05d}.avro" for k in range(3)]      # each prior model read once
    assert model.last_training_stats["nit"].max() <= 1                                          # warm start at the optimum
    for k in range(3):
        again = list(avro.read_file(str(tmp_path / "models" / f"part-{k:05d}.avro")))
        assert [r["modelId"] for r in again] == [r["modelId"] for r in first[k]]
        for a, b in zip(again, first[k]):
            np.testing.assert_allclose([c["value"] for c in a["means"]], [c["value"] for c in b["means"]], rtol=1e-6)


def test_driver_pipeline_raises_what_a_background_write_raised(tmp_path, monkeypatch):
    monkeypatch.delenv("TF_CONFIG", raising=False)
    base, model = _pipeline_job(tmp_path, parts=(0, 1))

    def boom(*a, **k):
        raise OSError("disk full")
    monkeypatch.setattr(model, "_save_model", boom)
    with pytest.raises(OSError, match="disk full"):
        RandomEffectDriver(base, model).run_training(SCHEMA)
    assert model._io_pool is None


def test_model_table_rows_round_trip():
    """rows_for / from_rows (what travels with a re-balanced entity): several chunks, an overridden id, ids without a model."""
    t = ModelTable()
    t.add_chunk(["a", "b", "c"], np.arange(6.0), [0, 3, 4, 6], np.array([5, 7, 2]), [0, 2, 2, 3])
    t.add_chunk(["b", "d"], np.array([10.0, 11.0, 12.0]), [0, 2, 3], np.array([9]), [0, 1, 1])
    ids = ["d", "zz", "b", "a"]
    rows = t.rows_for(ids)
    assert rows["has"].tolist() == [True, False, True, True]
    assert rows["coef_ptr"].tolist() == [0, 1, 1, 3, 6] and rows["feat_ptr"].tolist() == [0, 0, 0, 1, 3]
    assert rows["theta"].tolist() == [12.0, 10.0, 11.0, 0.0, 1.0, 2.0] and rows["idx"].tolist() == [9, 5, 7]
    back = ModelTable.from_rows(ids, rows)
    assert sorted(back.keys()) == ["a", "b", "d"] and "zz" not in back
    for k in ("a", "b", "d"):
        assert np.array_equal(back[k].theta, t[k].theta) and np.array_equal(back[k].unique_global_indices, t[k].unique_global_indices)
    empty = ModelTable().rows_for(["x"])
    assert not empty["has"].any() and empty["theta"].size == 0


def test_entity_id_listed_twice_is_scored_with_its_later_model(tmp_path):
    """The reference keeps one model per id (dict.update: the later record wins) and scores every record of that id with
    it. The scoring pass of the partition just trained on normally reuses the solve's own coefficients; with a repeated id
    it has to go through the model table instead — the scores must equal a cold prediction from the saved model file."""
    from gdmix_amd import synthetic
    b = synthetic.make_batch(6, 12, 4, 64, seed=3)
    b.entity_ids = ["a", "b", "a", "c", "b", "d"]
    md = {"features": [{"name": "bag", "dtype": "float", "shape": [64], "isSparse": True},
                       {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                       {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                       {"name": "ent", "dtype": "string", "shape": [], "isSparse": False}],
          "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}
    json.dump(md, open(tmp_path / "meta.json", "w"))
    with open(tmp_path / "features.csv", "w") as f:
        f.write("".join(f"f{i},\n" for i in range(64)))
    d = tmp_path / "train" / "partitionId=0"
    write_grouped_partition(str(d / "part-0.tfrecord"), b, "ent", "bag", weight_column_name=None)
    argv = ["--uid_column_name", "uid", "--label_column_name", "response", "--output_model_dir", str(tmp_path / "models"),
            "--metadata_file", str(tmp_path / "meta.json"), "--feature_bag", "bag", "--feature_file", str(tmp_path / "features.csv"),
            "--partition_entity", "ent", "--regularize_bias", "False"]
    schema = SchemaParams(uid_column_name="uid", label_column_name="response", prediction_score_column_name="predictionScore")
    m = RandomEffectLRLBFGSModel(argv)
    m._solver = OracleSolverDouble()
    m.train(str(d), None, m.metadata_file, None, {"partition_index": 0, "active_training_output_file": str(tmp_path / "ts.avro")}, schema)
    models = list(avro.read_file(str(tmp_path / "models" / "part-00000.avro")))
    assert [r["modelId"] for r in models] == ["a", "b", "c", "d"]               # first position kept, later values
    m2 = RandomEffectLRLBFGSModel(argv)
    m2._solver = OracleSolverDouble()
    m2.predict(str(tmp_path / "cold"), str(d), m2.metadata_file, str(tmp_path / "models"), {"partition_index": 0}, schema)
    cold = list(avro.read_file(str(tmp_path / "cold" / "part-00000.avro")))
    assert list(avro.read_file(str(tmp_path / "ts.avro"))) == cold and len(cold) == b.N
    # the first record of "a" is scored with the model trained on the second one
    n0 = int(b.ent_row_ptr[1])
    mean_a = {c["name"]: c["value"] for c in models[0]["means"]}
    k0, k1 = int(b.row_nnz_ptr[0]), int(b.row_nnz_ptr[1])
    z = mean_a["(INTERCEPT)"] + sum(float(b.val[k]) * mean_a.get(f"f{int(b.col_global[k])}", 0.0) for k in range(k0, k1)) + float(b.offset[0])
    assert abs(cold[0]["predictionScore"] - z) < 1e-5 and n0 > 0


# ---- the model's recovery paths (they had no test) -------------------------------------------------------------------

def test_an_aborted_team_barrier_is_solved_again_without_the_tall_team_class_and_the_setting_comes_back():
    """_pack_and_solve: GDMIX_RE_ST_ABORTED from a tall-team barrier -> the partition once more with every tall entity on one
    workgroup (set_tall_team_n(0)), then the solver's own threshold again — also when the second solve raises."""
    import types
    from gdmix_amd.model import RandomEffectLRLBFGSModel as M
    calls = []

    class Solved:
        def __init__(self, status):
            self.status = status

        def to_host(self, keys):
            return {"status": np.array(self.status, np.int32), "theta_thr": np.zeros(3)}

    class Solver:
        tall_team_n = 16384

        def set_tall_team_n(self, n):
            calls.append(("set", n))
            self.tall_team_n = n

        def solve(self, packed, opts, theta0=None):
            calls.append(("solve", self.tall_team_n))
            return Solved([0, 9, 1] if self.tall_team_n else [0, 0, 1])
    me = types.SimpleNamespace(ST_ABORTED=M.ST_ABORTED, _STAT_KEYS=M._STAT_KEYS)
    packs = []
    s = Solver()
    packed, solved, res = M._pack_and_solve(me, s, lambda: packs.append(1) or "packed", None, None)
    assert list(res["status"]) == [0, 0, 1] and len(packs) == 2
    assert calls == [("solve", 16384), ("set", 0), ("solve", 0), ("set", 16384)] and s.tall_team_n == 16384
    # a clean solve: one pack, nothing switched
    calls.clear()
    s2 = Solver()
    s2.tall_team_n = 0          # the class is off (GDMIX_RE_TALL_TEAM=0): an ABORTED from a team TIER is not retried, the caller sees it
    s2.solve = lambda packed, opts, theta0=None: Solved([9, 0, 0])
    _, _, res = M._pack_and_solve(me, s2, lambda: "packed", None, None)
    assert list(res["status"]) == [9, 0, 0] and calls == []
    # the retry itself fails: the threshold is restored all the same
    s3 = Solver()
    n = {"solve": 0}

    def flaky(packed, opts, theta0=None):
        n["solve"] += 1
        if n["solve"] == 2:
            raise RuntimeError("device lost")
        return Solved([9, 0, 0])
    s3.solve = flaky
    with pytest.raises(RuntimeError, match="device lost"):
        M._pack_and_solve(me, s3, lambda: "packed", None, None)
    assert s3.tall_team_n == 16384


def test_partitions_with_and_without_weights_do_not_go_into_one_device_batch():
    """_cat_wire: a group is one wire batch — partitions where only some carry a weight column cannot be one (the caller then solves
    each on its own); widths are widened to the widest, counts add up."""
    import torch
    from gdmix_amd.model import RandomEffectLRLBFGSModel as M
    from gdmix_amd.solver import REDeviceSolver

    def wire(E, N, Z, col_dtype, weight):
        w = {k: None for k in REDeviceSolver.WIRE_ARRAYS}
        w.update(E=E, N=N, Z=Z, row_nnz_width=1, col_width=torch.empty(0, dtype=col_dtype).element_size(), y_width=1,
                 ent_n=torch.full((E,), N // E, dtype=torch.int32), row_nnz=torch.ones(N, dtype=torch.uint8),
                 col_global=torch.arange(Z, dtype=col_dtype), val=torch.ones(Z), y=torch.zeros(N, dtype=torch.uint8),
                 offset=torch.zeros(N), weight=torch.ones(N) if weight else None)
        return w
    a, b = wire(2, 4, 4, torch.int16, True), wire(1, 3, 3, torch.int32, True)
    cat = M._cat_wire(torch, [a, b])
    assert (cat["E"], cat["N"], cat["Z"], cat["col_width"]) == (3, 7, 7, 4)
    assert cat["col_global"].dtype == torch.int32 and cat["col_global"].tolist() == [0, 1, 2, 3, 0, 1, 2]
    assert cat["weight"].numel() == 7
    assert M._cat_wire(torch, [a, wire(1, 3, 3, torch.int32, False)]) is None
    none = M._cat_wire(torch, [wire(2, 4, 4, torch.int16, False), wire(1, 3, 3, torch.int16, False)])
    assert none is not None and none["weight"] is None
