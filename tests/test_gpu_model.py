"""End-to-end on the MI355X: the RE model API and the CLI with the real device solver — TFRecord partitions
in, photon-ml Avro model and score Avro out — against the reference's golden coefficients."""
import json
import os
import shutil

import numpy as np
import pytest

from helpers import GOLDEN, load_fixture, well_posed_mask
from gdmix_amd import gdmix as cli
from gdmix_amd.io import avro
from gdmix_amd.io.grouped_reader import write_grouped_partition
from gdmix_amd.model import RandomEffectLRLBFGSModel
from gdmix_amd.params import SchemaParams

pytestmark = pytest.mark.gpu
RES = os.path.join(GOLDEN, "ref_resources")
SCHEMA = SchemaParams(uid_column_name="uid", weight_column_name="weight", label_column_name="response",
                      prediction_score_column_name="predictionScore")


def test_reference_fixture_through_model_api(tmp_path):
    d = tmp_path / "train" / "partitionId=0"
    os.makedirs(d)
    shutil.copy(os.path.join(RES, "data.tfrecord"), d / "data.tfrecord")
    argv = ["--uid_column_name", "uid", "--weight_column_name", "weight", "--label_column_name", "response",
            "--output_model_dir", str(tmp_path / "models"), "--metadata_file", os.path.join(RES, "data.json"),
            "--feature_bag", "per_member", "--feature_file", os.path.join(RES, "fake_feature_file.csv"),
            "--partition_entity", "memberId", "--l2_reg_weight", "0.1"]
    m = RandomEffectLRLBFGSModel(argv)
    ctx = {"partition_index": 0, "active_training_output_file": str(tmp_path / "ts" / "a.avro")}
    m.train(str(d), None, m.metadata_file, str(tmp_path / "models"), ctx, SCHEMA)
    recs = list(avro.read_file(str(tmp_path / "models" / "part-00000.avro")))
    _, _, exp, _ = load_fixture("ref_fixture_l2_0.1")
    np.testing.assert_allclose([c["value"] for c in recs[0]["means"]], exp["theta_thr"][:8], rtol=1e-7)
    np.testing.assert_allclose([c["value"] for c in recs[1]["means"]], exp["theta_thr"][8:], rtol=1e-7)
    assert np.array_equal(m.last_training_stats["nit"], exp["nit"])
    scores = list(avro.read_file(str(tmp_path / "ts" / "a.avro")))
    m2 = RandomEffectLRLBFGSModel(argv)
    m2.predict(str(tmp_path / "cold"), str(d), m2.metadata_file, str(tmp_path / "models"), {"partition_index": 0}, SCHEMA)
    assert list(avro.read_file(str(tmp_path / "cold" / "part-00000.avro"))) == scores


def test_cli_train_then_inference_on_partitioned_c2_data(tmp_path):
    """python -m gdmix_amd.gdmix --stage=random_effect: 3 partitions of C2-shaped entities, shipped MovieLens
    options, coefficients vs the reference fixture; then --action=inference reproduces the training scores."""
    b, opts, exp, _ = load_fixture("c2_shipped_cfg")
    md = {"features": [{"name": "bag", "dtype": "float", "shape": [1024], "isSparse": True},
                       {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                       {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                       {"name": "ent", "dtype": "string", "shape": [], "isSparse": False}],
          "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}
    json.dump(md, open(tmp_path / "meta.json", "w"))
    with open(tmp_path / "features.csv", "w") as f:
        f.write("".join(f"f{i},\n" for i in range(1024)))
    parts = [0, 1, 2]
    per = b.E // 3
    for k in parts:
        sub = b.select(np.arange(k * per, (k + 1) * per))
        write_grouped_partition(str(tmp_path / "train" / "active" / f"partitionId={k}" / "part-0.tfrecord.gz"), sub,
                                "ent", "bag", weight_column_name=None)
        write_grouped_partition(str(tmp_path / "valid" / f"partitionId={k}" / "part-0.tfrecord"), sub, "ent", "bag",
                                weight_column_name=None)
    open(tmp_path / "plist.txt", "w").write("0,1,2")
    common = ["gdmix", "--stage=random_effect", "--model_type=logistic_regression", "--uid_column_name=uid",
              "--label_column_name=response", "--prediction_score_column_name=predictionScore",
              f"--partition_list_file={tmp_path / 'plist.txt'}", f"--training_data_dir={tmp_path / 'train'}",
              f"--validation_data_dir={tmp_path / 'valid'}", f"--metadata_file={tmp_path / 'meta.json'}",
              f"--output_model_dir={tmp_path / 'models'}", "--feature_bag=bag", f"--feature_file={tmp_path / 'features.csv'}",
              "--partition_entity=ent", "--regularize_bias=False", "--l2_reg_weight=1.0",
              f"--training_score_dir={tmp_path / 'ts'}", f"--validation_score_dir={tmp_path / 'vs'}"]
    os.environ.pop("TF_CONFIG", None)
    cli.run(common + ["--action=train"])
    wp = well_posed_mask(b, opts)
    ic_ptr = exp["ent_feat_ptr"] + np.arange(b.E + 1)
    for k in parts:
        recs = list(avro.read_file(str(tmp_path / "models" / f"part-{k:05d}.avro")))
        assert len(recs) == per
        for r_i, rec in enumerate(recs):
            e = k * per + r_i
            assert rec["modelId"] == b.entity_ids[e]
            if not wp[e]:
                continue
            want = exp["theta_thr"][ic_ptr[e]:ic_ptr[e + 1]]
            keep = np.concatenate([[True], np.abs(want[1:]) > 1e-4])
            np.testing.assert_allclose([c["value"] for c in rec["means"]], want[keep], rtol=1e-7)
            assert [c["name"] for c in rec["means"][1:]] == [f"f{g}" for g in exp["unique_global"][exp["ent_feat_ptr"][e]:exp["ent_feat_ptr"][e + 1]][keep[1:]]]
    train_scores = {k: list(avro.read_file(str(tmp_path / "ts" / f"partitionId={k}" / "part-00000-active.avro"))) for k in parts}
    valid_scores = {k: list(avro.read_file(str(tmp_path / "vs" / f"partitionId={k}" / "part-00000.avro"))) for k in parts}
    assert train_scores == valid_scores
    infer = [a for a in common if not a.startswith("--training_score_dir") and not a.startswith("--validation_score_dir")]
    cli.run(infer + ["--action=inference", f"--validation_score_dir={tmp_path / 'vs2'}"])
    for k in parts:
        again = list(avro.read_file(str(tmp_path / "vs2" / f"partitionId={k}" / f"part-{k:05d}.avro")))
        assert again == valid_scores[k]


def _means_by_feature(rec):
    """modelId, {global feature index: value} (intercept under -1) of one BayesianLinearModelAvro record."""
    out = {}
    for c in rec["means"]:
        out[-1 if c["name"] == "(INTERCEPT)" else int(c["name"][1:])] = c["value"]
    return out


@pytest.mark.parametrize("shape", ["c5", "ml20m_movie"])
def test_cli_process_on_zipf_and_movielens_shaped_partition_directories(tmp_path, shape):
    """The real pipeline of BASELINE configs[2] and [4] at reduced size: a partition directory whose entities were hashed into
    partitions the reference's way (io/input_data_pipeline.py:223-332 reads it, util/io_utils.py:163-212 is the model file) —
    C5: Zipf sizes of SURVEY 8(d) plus one entity above 2^17 non-zeros (team tiers); MovieLens-20M per-movie: tall and skinny,
    a third of the entities with one rating — trained by a real `python -m gdmix_amd.gdmix` child process. A sample stratified by
    entity size is compared with the oracle: every model present once, thresholded zero pattern and coefficients."""
    import subprocess
    import sys
    from gdmix_amd import synthetic
    from gdmix_amd.batch import concat
    from gdmix_amd.partition_dirs import write_partition_dir
    from helpers import oracle_solve_parallel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shape == "c5":
        b = concat([synthetic.make_survey_batch(6000, 32, 8, 65536, seed=41, size_dist="c5zipf", with_uid=True),
                    synthetic.make_survey_batch(1, 17000, 8, 65536, seed=42, size_dist="const", entity_id_base=900_000, with_uid=True)])
        b.uid = np.arange(b.N, dtype=np.int64)
        dim = 65536
        assert b.ent_nnz().max() >= (1 << 17)
    else:
        b = synthetic.make_movielens_20m("per_movie", seed=43, entities=2500, with_uid=True)
        dim = 24
        assert b.ent_n().max() > 4096 and (b.ent_n() == 1).sum() > 100
    argv, members, _ = write_partition_dir(str(tmp_path), b, 4, dim)
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("TF_CONFIG", None)
    r = subprocess.run([sys.executable, "-m", "gdmix_amd.gdmix"] + argv[1:], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    models = {}
    for k, own in members.items():
        recs = list(avro.read_file(str(tmp_path / "models" / f"part-{k:05d}.avro")))
        assert [rec["modelId"] for rec in recs] == [b.entity_ids[e] for e in own]       # file order = partition order, every entity once
        for e, rec in zip(own, recs):
            models[int(e)] = rec
    assert len(models) == b.E
    scores = sum(len(list(avro.read_file(str(tmp_path / "ts" / f"partitionId={k}" / "part-00000-active.avro")))) for k in members)
    assert scores == b.N
    # the sample: the largest entities, the smallest, and a random draw of the rest
    nnz = b.ent_nnz()
    order = np.argsort(-nnz, kind="stable")
    rng = np.random.default_rng(7)
    sample = np.unique(np.concatenate([order[:6], order[-20:], rng.choice(b.E, 150, replace=False)]))
    sub = b.select(sample)
    kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
    pk, ref = oracle_solve_parallel(sub, kw)
    y1 = np.add.reduceat(sub.y.astype(np.float64), sub.ent_row_ptr[:-1])
    wp = (y1 > 0) & (y1 < sub.ent_n())
    cp = pk["ent_feat_ptr"] + np.arange(sub.E + 1)
    worst, compared = 0.0, 0
    for i, e in enumerate(sample):
        got = _means_by_feature(models[int(e)])
        want = ref["theta_thr"][cp[i]:cp[i + 1]]
        feats = pk["unique_global"][pk["ent_feat_ptr"][i]:pk["ent_feat_ptr"][i + 1]]
        assert -1 in got                                                                   # the intercept is always written (io_utils.py:118)
        if not wp[i]:
            # class D (all labels equal, unregularised intercept): no finite optimum; every other coefficient is thresholded away
            assert set(got) == {-1} and np.sign(got[-1]) == (1.0 if y1[i] > 0 else -1.0)
            continue
        live = {int(f): v for f, v in zip(feats, want[1:]) if abs(v) > 1e-4}
        # a coefficient within rounding of the threshold may fall on either side
        edge = {int(f) for f, v in zip(feats, want[1:]) if abs(abs(v) - 1e-4) < 1e-9}
        assert (set(got) - {-1}) ^ set(live) <= edge, (int(e), sorted((set(got) - {-1}) ^ set(live))[:5])
        scale = max(np.max(np.abs(want)), 1e-300)
        err = max([abs(got[-1] - want[0])] + [abs(got[f] - v) for f, v in live.items() if f in got]) / scale
        worst = max(worst, err)
        compared += 1
    assert compared >= 100 and worst <= 1e-5, (compared, worst)      # north_star: coefficients within 1e-5 rel-err of the reference L-BFGS


@pytest.mark.parametrize("dim,k", [(1024, 16), (1 << 17, 512)])
def test_a_partition_the_reader_narrowed_packs_to_the_same_arrays(tmp_path, dim, k):
    """native reader with wire=True (gdmix_io_narrow) -> upload of the 32-bit form -> gdmix_re_widen -> pack: the packed batch and the
    solve are bit for bit those of the 64-bit hand-over, for both widths of the counts and the indices."""
    from gdmix_amd import synthetic
    from gdmix_amd.batch import WireRawBatch
    from gdmix_amd.io.grouped_reader import read_grouped_partition
    from gdmix_amd.solver import REDeviceSolver, SolverOptions
    b = synthetic.make_batch(300, 12, k, dim, seed=5)
    md = {"features": [{"name": "bag", "dtype": "float", "shape": [dim], "isSparse": True},
                       {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                       {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                       {"name": "ent", "dtype": "string", "shape": [], "isSparse": False}],
          "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}
    write_grouped_partition(str(tmp_path / "part-0.tfrecord"), b, "ent", "bag", "offset", "uid", "response", None)
    args = (str(tmp_path), md, "ent", "bag", "offset", "uid", "response", None)
    wide = read_grouped_partition(*args, num_features=dim, native=True)
    narrow = read_grouped_partition(*args, num_features=dim, native=True, wire=True)
    assert isinstance(narrow, WireRawBatch) and not isinstance(wide, WireRawBatch)
    assert narrow.to_wire()["col_width"] == (2 if dim <= 65536 else 4) and narrow.to_wire()["row_nnz_width"] == (1 if k <= 255 else 2)
    s = REDeviceSolver(0)
    pa, pb = s.pack(wide), s.pack(narrow)
    assert narrow._row_nnz_ptr is None and narrow._col_global is None       # the 64-bit arrays were never rebuilt on the host
    for name in ("ent_feat_ptr", "unique_global", "ent_nnz_ptr"):
        assert np.array_equal(getattr(pa, name)().cpu().numpy(), getattr(pb, name)().cpu().numpy()), name
    o = SolverOptions(l2=1.0, has_intercept=True, m=10, max_iter=100, ftol=1e-12)
    ra, rb = s.solve(pa, o).to_host(), s.solve(pb, o).to_host()
    for key in ("theta", "nit", "nfev", "status", "fval"):
        assert np.array_equal(ra[key], rb[key]), key
    s.close()


def test_partitions_solved_in_one_device_batch_give_the_files_of_partition_by_partition(tmp_path, monkeypatch, caplog):
    """model.plan_group: the consecutive cold partitions of a run are solved in ONE device batch (their wire forms concatenated in HBM)
    and every partition still gets its own model and score files. Six partitions of Zipf-sized entities — one directory empty, one with
    a prior model (it trains on its own, warm) — once partition by partition (GROUP_MAX = 1), once grouped: the same entities in the
    same files, coefficients to 1e-7 of their scale (an entity's kernel is chosen per batch: its sums may be ordered differently, not
    more), scores to 4e-6; the warm-started partition identical (it is not in a group either way)."""
    import logging
    from gdmix_amd import model as model_mod, synthetic
    from gdmix_amd.partition_dirs import write_partition_dir
    b = synthetic.make_survey_batch(3000, 24, 6, 4096, seed=77, size_dist="c5zipf", with_uid=True)
    b.uid = np.arange(b.N, dtype=np.int64)
    runs = {}
    for name, limit in (("single", 1), ("grouped", 8)):
        root = tmp_path / name
        os.makedirs(root)
        argv, members, _ = write_partition_dir(str(root), b, 6, 4096)
        assert sorted(members) == [0, 1, 2, 3, 4, 5]
        # partition 2: an empty directory in the middle of the list; partition 4: a prior model (trained here, alone, first)
        empty = root / "train" / "active" / "partitionId=2"
        shutil.rmtree(empty)
        os.makedirs(empty)
        monkeypatch.setattr(model_mod, "GROUP_MAX", 1)
        os.environ.pop("TF_CONFIG", None)
        open(root / "plist.txt", "w").write("4")
        cli.run(argv)
        prior = list(avro.read_file(str(root / "models" / "part-00004.avro")))
        open(root / "plist.txt", "w").write("0,1,2,3,4,5")
        monkeypatch.setattr(model_mod, "GROUP_MAX", limit)
        caplog.clear()
        with caplog.at_level(logging.INFO, logger="gdmix_amd.model"):
            cli.run(argv)
        said = [r.getMessage() for r in caplog.records if "solved in one device batch" in r.getMessage()]
        assert (said == []) if limit == 1 else (len(said) == 1 and said[0].startswith("3 partitions")), said
        models = {k: list(avro.read_file(str(root / "models" / f"part-{k:05d}.avro"))) for k in members if k != 2}
        scores = {k: list(avro.read_file(str(root / "ts" / f"partitionId={k}" / "part-00000-active.avro"))) for k in members if k != 2}
        assert not os.path.exists(root / "models" / "part-00002.avro")
        assert len(models[4]) == len(prior)
        runs[name] = (models, scores)
    (m1, s1), (m2, s2) = runs["single"], runs["grouped"]
    assert m1[4] == m2[4] and s1[4] == s2[4]
    worst = 0.0
    for k in m1:
        assert [r["modelId"] for r in m1[k]] == [b.entity_ids[e] for e in members[k]] == [r["modelId"] for r in m2[k]]
        for r1, r2 in zip(m1[k], m2[k]):
            a, c = _means_by_feature(r1), _means_by_feature(r2)
            scale = max(max(abs(v) for v in a.values()), 1e-300)
            if abs(a[-1]) > 12:         # class D (all labels equal): the unregularised intercept runs off, chaotically; nothing else is kept
                assert set(a) == set(c) == {-1} and np.sign(a[-1]) == np.sign(c[-1])
                continue
            # a coefficient within rounding of the export threshold may be kept in one run and dropped in the other
            assert all(abs(abs((a.get(f) or c.get(f))) - 1e-4) < 1e-9 for f in set(a) ^ set(c)), (k, r1["modelId"])
            worst = max(worst, max(abs(a[f] - c[f]) for f in set(a) & set(c)) / scale)
        assert [r["uid"] for r in s1[k]] == [r["uid"] for r in s2[k]]
        p1 = np.array([r["predictionScore"] for r in s1[k]], np.float32)
        p2 = np.array([r["predictionScore"] for r in s2[k]], np.float32)
        sane = np.abs(p1) < 12
        assert np.all(np.abs(p1[sane] - p2[sane]) <= 4e-6)
    assert worst <= 1e-7, worst
