"""FixedEffectLRModelLBFGS / FixedEffectDriver / `--stage=fixed_effect` (gdmix_amd/fe_model.py): files in, files out,
against the reference's own expected coefficients and scores (tests/golden/fe_*.npz). CPU tests run the model class over
the oracle-backed double; the GPU test runs the CLI end to end on the device."""
import json
import os

import numpy as np
import pytest

from gdmix_amd import constants
from gdmix_amd.driver import FixedEffectDriver
from gdmix_amd.fe_model import FixedEffectLRModelLBFGS, shard_input_files
from gdmix_amd.io import avro, tfrecord
from gdmix_amd.params import Params, SchemaParams
from helpers import OracleFeDouble

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    z = np.load(os.path.join(HERE, "golden", f"fe_{name}.npz"))
    return {k: z[k] for k in z.files}


def write_examples(path, rp, col, val, y, off, uid0, bag="global", linear=False, with_weight=False):
    recs = []
    for i in range(rp.size - 1):
        a, b = int(rp[i]), int(rp[i + 1])
        f = {"uid": ("int64", [uid0 + i]), "offset": ("float", [float(off[i])]),
             "response": ("float", [float(y[i])]) if linear else ("int64", [int(y[i])])}
        if bag:
            f[f"{bag}_indices"] = ("int64", col[a:b])
            f[f"{bag}_values"] = ("float", val[a:b])
        if with_weight:
            f["weight"] = ("float", [2.5])
        recs.append(tfrecord.encode_example(f))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tfrecord.write_records(path, recs)


def setup_case(tmp_path, c, n_files=3, bag="global", with_weight=False):
    D = int(c["num_features"])
    linear = bool(c["linear"])
    n = c["y"].size
    cuts = np.linspace(0, n, n_files + 1).astype(int)
    for name, rp, col, val, y, off, base in (("train", c["row_nnz_ptr"], c["col_global"], c["val"], c["y"], c["offset"], 0),
                                             ("valid", c["v_row_nnz_ptr"], c["v_col_global"], c["v_val"], c["v_y"], c["v_offset"], 10_000)):
        for f in range(n_files):
            r0, r1 = cuts[f], cuts[f + 1]
            sub = rp[r0:r1 + 1] - rp[r0]
            write_examples(str(tmp_path / name / f"part-{f:05d}.tfrecord"), sub, col[rp[r0]:rp[r1]], val[rp[r0]:rp[r1]], y[r0:r1],
                           off[r0:r1], base + r0, bag=bag if D else None, linear=linear, with_weight=with_weight)
    feats = [{"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
             {"name": "offset", "dtype": "float", "shape": [], "isSparse": False}]
    if D:
        feats.append({"name": bag, "dtype": "float", "shape": [D], "isSparse": True})
    if with_weight:
        feats.append({"name": "weight", "dtype": "float", "shape": [], "isSparse": False})
    md = {"features": feats, "labels": [{"name": "response", "dtype": "float" if linear else "int", "shape": [], "isSparse": False}]}
    json.dump(md, open(tmp_path / "meta.json", "w"))
    with open(tmp_path / "features.csv", "w") as f:
        f.write("".join(f"f{i},t{i % 3}\n" for i in range(D)))
    argv = ["gdmix", "--stage=fixed_effect", "--action=train", f"--model_type={'linear_regression' if linear else 'logistic_regression'}",
            "--uid_column_name=uid", "--label_column_name=response", "--prediction_score_column_name=predictionScore",
            f"--training_data_dir={tmp_path / 'train'}", f"--validation_data_dir={tmp_path / 'valid'}", f"--metadata_file={tmp_path / 'meta.json'}",
            f"--output_model_dir={tmp_path / 'model'}", f"--training_score_dir={tmp_path / 'ts'}", f"--validation_score_dir={tmp_path / 'vs'}",
            f"--l2_reg_weight={float(c['l2'])}", f"--has_intercept={bool(c['has_intercept'])}", f"--regularize_bias={bool(c['has_intercept'])}",
            f"--num_of_lbfgs_iterations={int(c['max_iter'])}"]
    if D:
        argv += [f"--feature_bag={bag}", f"--feature_file={tmp_path / 'features.csv'}"]
    if with_weight:
        argv += ["--weight_column_name=weight"]
    for d in ("model", "ts", "vs"):
        os.makedirs(tmp_path / d, exist_ok=True)
    return argv


def check_outputs(tmp_path, c, score_tol=2e-6):
    D = int(c["num_features"])
    ic = int(c["has_intercept"])
    theta = c["theta"]
    recs = list(avro.read_file(str(tmp_path / "model" / "part-00000.avro")))
    assert len(recs) == 1 and recs[0]["modelId"] == "global model"
    means = {(m["name"], m["term"]): m["value"] for m in recs[0]["means"]}
    if ic:
        assert recs[0]["means"][0]["name"] == constants.INTERCEPT
        np.testing.assert_allclose(means[(constants.INTERCEPT, "")], theta[-1], rtol=1e-7)
    for j in range(D):
        key = (f"f{j}", f"t{j % 3}")
        if abs(theta[j]) > 1e-4:
            np.testing.assert_allclose(means[key], theta[j], rtol=1e-7)
        else:
            assert key not in means
    linear = bool(c["linear"])
    for sub, n_exp, per, tot, base, has_file in (("ts", c["y"].size, c["train_per_coord"], c["train_score"], 0, not linear),
                                                 ("vs", c["v_y"].size, c["valid_per_coord"], c["valid_score"], 10_000, True)):
        path = tmp_path / sub / "part-00000.avro"
        if not has_file:
            assert not path.exists()   # no scoring of the training data for plain linear regression
            continue
        rows = list(avro.read_file(str(path)))
        assert len(rows) == n_exp
        got = {r["uid"]: r for r in rows}
        for i in range(n_exp):
            r = got[base + i]
            # scores of the thresholded model: bounded by the coefficients dropped (<= 1e-4 each)
            assert abs(r["predictionScore"] - tot[i]) <= 1e-3 and abs(r["predictionScorePerCoordinate"] - per[i]) <= 1e-3
        assert set(rows[0]) >= {"uid", "predictionScore", "response", "predictionScorePerCoordinate"}


@pytest.mark.parametrize("name", ["logistic_offset", "linear_offset", "logistic_no_intercept", "logistic_intercept_only", "logistic_l2_0.01"])
def test_train_writes_model_and_scores_matching_reference_expectations(tmp_path, name):
    c = load(name)
    argv = setup_case(tmp_path, c)
    params = Params.__from_argv__(argv, error_on_unknown=False)
    model = FixedEffectLRModelLBFGS(argv, params)
    model._fe = OracleFeDouble()
    driver = FixedEffectDriver(params, model)
    driver.run_training(SchemaParams.__from_argv__(argv, error_on_unknown=False), export_model=True)
    assert model.last_training_info["status"] in (0, 1)
    check_outputs(tmp_path, c)
    # inference from the saved model reproduces the validation scores
    os.remove(tmp_path / "vs" / "part-00000.avro")
    driver.run_inference(SchemaParams.__from_argv__(argv, error_on_unknown=False))
    rows = list(avro.read_file(str(tmp_path / "vs" / "part-00000.avro")))
    assert len(rows) == c["v_y"].size


def test_warm_start_from_the_saved_model_is_a_fixed_point(tmp_path):
    c = load("logistic_offset")
    argv = setup_case(tmp_path, c)
    params = Params.__from_argv__(argv, error_on_unknown=False)
    sp = SchemaParams.__from_argv__(argv, error_on_unknown=False)
    m1 = FixedEffectLRModelLBFGS(argv, params)
    m1._fe = OracleFeDouble()
    FixedEffectDriver(params, m1).run_training(sp)
    first = m1.model_coefficients.copy()
    m2 = FixedEffectLRModelLBFGS(argv, params)
    m2._fe = OracleFeDouble()
    FixedEffectDriver(params, m2).run_training(sp)
    assert m2.last_training_info["nit"] <= 1                      # starts at the optimum
    np.testing.assert_allclose(m2.model_coefficients, first, rtol=0, atol=2e-4)


def test_weights_truncate_in_the_score_file_as_in_the_reference(tmp_path):
    c = load("logistic_offset")
    argv = setup_case(tmp_path, c, with_weight=True)
    params = Params.__from_argv__(argv, error_on_unknown=False)
    model = FixedEffectLRModelLBFGS(argv, params)
    model._fe = OracleFeDouble()
    FixedEffectDriver(params, model).run_training(SchemaParams.__from_argv__(argv, error_on_unknown=False))
    rows = list(avro.read_file(str(tmp_path / "ts" / "part-00000.avro")))
    assert all(r["weight"] == 2.0 for r in rows)                  # int(2.5), fixed_effect_lr_lbfgs_model.py:427-428


def test_native_example_reader_matches_python_decoder(tmp_path):
    from gdmix_amd.fe_model import read_per_record_files
    from gdmix_amd.io.metadata import DatasetMetadata
    c = load("logistic_wide")
    setup_case(tmp_path, c, with_weight=True)
    md = DatasetMetadata(str(tmp_path / "meta.json"))
    files = sorted(str(p) for p in (tmp_path / "train").iterdir())
    args = (files, md, "global", int(c["num_features"]), "uid", "response", "offset", "weight")
    a = read_per_record_files(*args, native=False)
    b = read_per_record_files(*args, native=True)
    assert a["n"] == b["n"] == c["y"].size and a["has_label"] == b["has_label"] and a["has_weight"] == b["has_weight"]
    for k in ("row_nnz_ptr", "col", "val", "y", "offset", "weight", "uid"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    np.testing.assert_array_equal(b["col"], c["col_global"])
    # columns the metadata does not list fall back to defaults in both
    md2 = json.load(open(tmp_path / "meta.json"))
    md2["features"] = [f for f in md2["features"] if f["name"] not in ("offset", "weight")]
    json.dump(md2, open(tmp_path / "meta2.json", "w"))
    for native in (False, True):
        r = read_per_record_files(files, DatasetMetadata(str(tmp_path / "meta2.json")), "global", int(c["num_features"]), "uid",
                                  "response", "offset", "weight", native=native)
        assert not r["offset"].any() and (r["weight"] == 1).all() and not r["has_weight"]
    # a feature index outside the bag fails in both
    for native in (False, True):
        with pytest.raises(ValueError):
            read_per_record_files(files, md, "global", 5, "uid", "response", "offset", "weight", native=native)


def test_file_sharding_rules(tmp_path):
    for i in range(5):
        open(tmp_path / f"part-{i}.tfrecord", "wb").close()
    files = sorted(str(p) for p in tmp_path.iterdir())
    assert shard_input_files(str(tmp_path), 2, 0) == files[0::2] and shard_input_files(str(tmp_path), 2, 1) == files[1::2]
    assert shard_input_files(str(tmp_path), 8, 3) == [files[3]] and shard_input_files(str(tmp_path), 8, 6) == []
    assert shard_input_files(str(tmp_path / "*.tfrecord"), 1, 0) == files
    # the reference shards over every entry of the directory (glob '*'), whatever its suffix (distribution_utils.py:31-36)
    open(tmp_path / "_SUCCESS", "wb").close()
    open(tmp_path / "part-9.tfrecords", "wb").close()
    allf = sorted(str(p) for p in tmp_path.iterdir())
    assert len(allf) == 7 and shard_input_files(str(tmp_path), 2, 0) == allf[0::2] and shard_input_files(str(tmp_path), 2, 1) == allf[1::2]
    # ... sub-directories included (low_rpc_call_glob returns them): they keep their place in the stride and hold no records
    os.makedirs(tmp_path / "a_subdir")
    with_dir = sorted(str(p) for p in tmp_path.iterdir())
    assert str(tmp_path / "a_subdir") in with_dir and len(with_dir) == 8
    assert shard_input_files(str(tmp_path), 3, 1) == with_dir[1::3] and shard_input_files(str(tmp_path), 3, 0) == with_dir[0::3]


def _dense_variances(c, theta, mode, l2, regularize_bias):
    """The statement the reference's own test checks against (test_optimizer_helper.compute_coefficients_and_variance:
    -1 / diag(Hessian) resp. -diag(Hessian^-1) of the log-likelihood), in dense numpy, plus the l2 terms of :451-463."""
    D = int(c["num_features"])
    rp, col, val = c["row_nnz_ptr"], c["col_global"], c["val"].astype(np.float64)
    n = rp.size - 1
    X = np.zeros((n, D + 1))
    X[np.repeat(np.arange(n), np.diff(rp)), col] = val
    X[:, D] = 1.0
    z = X @ theta + c["offset"].astype(np.float64)
    rho = 1 / (1 + np.exp(-z))
    H = X.T @ (X * (rho * (1 - rho))[:, None])
    if mode == "simple":
        h = np.diagonal(H) + l2
        if not regularize_bias:
            h[-1] -= l2
        return 1.0 / (h + 1e-12)
    H = H + np.diag([l2 + 1e-12] * (D + 1))
    if not regularize_bias:
        H[-1, -1] -= l2
    return np.diagonal(np.linalg.inv(H))


@pytest.mark.parametrize("mode", ["simple", "full"])
def test_variance_modes_write_the_variances_of_the_thresholded_model(tmp_path, mode):
    """fixed_effect_variance_mode (fixed_effect_lr_lbfgs_model.py:271-305,451-463): variances of the saved (thresholded)
    coefficients on the training data, next to the means in the model file; the training scores are written even with
    scoring after training disabled, as upstream's variance pass rides on that scoring pass."""
    c = load("logistic_offset")
    argv = setup_case(tmp_path, c) + [f"--fixed_effect_variance_mode={mode}", "--disable_fixed_effect_scoring_after_training=True"]
    params = Params.__from_argv__(argv, error_on_unknown=False)
    model = FixedEffectLRModelLBFGS(argv, params)
    model._fe = OracleFeDouble()
    FixedEffectDriver(params, model).run_training(SchemaParams.__from_argv__(argv, error_on_unknown=False), export_model=True)
    assert (tmp_path / "ts" / "part-00000.avro").exists()
    D = int(c["num_features"])
    theta = model.model_coefficients
    want = _dense_variances(c, theta, mode, float(c["l2"]), bool(c["has_intercept"]))
    np.testing.assert_allclose(model.variances, want, rtol=1e-8)
    rec = list(avro.read_file(str(tmp_path / "model" / "part-00000.avro")))[0]
    means = {(m["name"], m["term"]): m["value"] for m in rec["means"]}
    var = {(m["name"], m["term"]): m["value"] for m in rec["variances"]}
    assert set(var) == set(means)                       # a variance for every coefficient written (gen_one_avro_model)
    np.testing.assert_allclose(var[(constants.INTERCEPT, "")], want[D], rtol=1e-8)
    for j in range(D):
        if abs(theta[j]) > 1e-4:
            np.testing.assert_allclose(var[(f"f{j}", f"t{j % 3}")], want[j], rtol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["logistic_offset", "linear_offset", "logistic_intercept_only", "logistic_wide"])
def test_cli_fixed_effect_stage_on_the_device(tmp_path, name):
    from gdmix_amd import gdmix
    c = load(name)
    argv = setup_case(tmp_path, c, n_files=2)
    gdmix.run(argv)
    check_outputs(tmp_path, c)
    inf = [a for a in argv if not a.startswith("--action")] + ["--action=inference"]
    os.remove(tmp_path / "vs" / "part-00000.avro")
    gdmix.run(inf)
    assert len(list(avro.read_file(str(tmp_path / "vs" / "part-00000.avro")))) == c["v_y"].size


def test_model_file_loads_the_same_through_the_native_reader_and_the_python_decoder(tmp_path, monkeypatch):
    from gdmix_amd.io import avro, native_reader
    c = load("logistic_offset")
    argv = setup_case(tmp_path, c)
    params = Params.__from_argv__(argv, error_on_unknown=False)
    sp = SchemaParams.__from_argv__(argv, error_on_unknown=False)
    m = FixedEffectLRModelLBFGS(argv, params)
    m._fe = OracleFeDouble()
    FixedEffectDriver(params, m).run_training(sp)
    called = []
    real = native_reader.read_models_avro
    monkeypatch.setattr(native_reader, "read_models_avro", lambda *a, **k: called.append(1) or real(*a, **k))
    nat = m._load_model()
    assert called
    monkeypatch.setattr(native_reader, "available", lambda: False)
    py = m._load_model()
    monkeypatch.undo()
    np.testing.assert_array_equal(nat, py)
    assert np.count_nonzero(nat) > 3 and nat[-1] != 0.0           # intercept last
    # a model naming a feature this job does not know: the reference skips it; the native reader declines, the decoder reads on
    path = [p for p in os.listdir(m.checkpoint_path) if p.endswith(".avro")][0]
    recs = list(avro.read_file(os.path.join(m.checkpoint_path, path)))
    recs[0]["means"].append({"name": "not_in_the_feature_file", "term": "", "value": 3.0})
    avro.write_file(os.path.join(m.checkpoint_path, path), avro.BAYESIAN_LINEAR_MODEL_SCHEMA, recs)
    np.testing.assert_array_equal(m._load_model(), py)
