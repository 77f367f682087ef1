#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own Python in the build container.

Runs only where /root/reference exists (never on the GPU box, never from tests). It imports
  gdmix.models.custom.scipy.job_consumers   (prepare_jobs, TrainingJobConsumer, InferenceJobConsumer)
  gdmix.models.custom.binary_logistic_regression.BinaryLogisticRegressionTrainer
from /root/reference/gdmix-trainer/src with `tensorflow` and `fastavro` replaced by empty stub
modules (they are only touched at import time on this path) and `dataset_reader` patched to identity,
feeds hand-built SparseTensorValue batches (the layout documented in job_consumers.py:176-199), and
records for every entity: the inputs, fmin_l_bfgs_b's raw result (theta, f, nit, funcalls, task,
grad), the thresholded theta, unique_global_indices and the optional variance.

Only data (inputs and expected outputs) is written; no reference source is copied.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_fixtures.py
"""
import collections
import collections.abc
import json
import os
import queue
import sys
import types
from types import SimpleNamespace

import numpy as np

REF = "/root/reference/gdmix-trainer/src"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

collections.Mapping = collections.abc.Mapping  # io_utils.namedtuple_with_defaults on py3.10
for _m in ("tensorflow", "fastavro"):
    sys.modules[_m] = types.ModuleType(_m)

import scipy  # noqa: E402
import gdmix.models.custom.scipy.job_consumers as jc  # noqa: E402
from gdmix.models.custom.binary_logistic_regression import BinaryLogisticRegressionTrainer  # noqa: E402

from gdmix_amd import synthetic  # noqa: E402
from gdmix_amd.batch import RawBatch, concat  # noqa: E402

jc.dataset_reader = lambda it: it
STV = collections.namedtuple("SparseTensorValue", ["indices", "values", "dense_shape"])

STATUS = {"CONVERGENCE: NORM OF PROJECTED GRADIENT <= PGTOL": 0,
          "CONVERGENCE: RELATIVE REDUCTION OF F <= FACTR*EPSMCH": 1,
          "STOP: TOTAL NO. OF ITERATIONS REACHED LIMIT": 2,
          "STOP: TOTAL NO. OF F,G EVALUATIONS EXCEEDS LIMIT": 3,
          "ABNORMAL ": 4}


def status_code(task):
    if isinstance(task, bytes):
        task = task.decode()
    t = task.strip().upper().replace("_", " ")
    for k, v in STATUS.items():
        if t.startswith(k.strip().upper().replace("_", " ")):
            return v
    if "PROJECTED GRADIENT" in t:
        return 0
    if "REDUCTION OF F" in t:
        return 1
    if "ITERATIONS REACHED" in t:
        return 2
    if "EVALUATIONS EXCEEDS" in t:
        return 3
    if "ABNORMAL" in t:
        return 4
    raise ValueError(f"unknown task {task!r}")


def tf_batches(b: RawBatch, batch_size, bag="bag", int_entity_ids=False):
    """Yield (features_val, labels_val) exactly as dataset_reader would for per_entity_grouped_input_fn."""
    n = b.ent_n()
    for e0 in range(0, b.E, batch_size):
        e1 = min(b.E, e0 + batch_size)
        f_idx, f_val, v_val = [], [], []
        d_idx, y_val, o_val, w_val, u_val = [], [], [], [], []
        for le, e in enumerate(range(e0, e1)):
            r0 = b.ent_row_ptr[e]
            for i in range(n[e]):
                d_idx.append((le, i))
                z0, z1 = b.row_nnz_ptr[r0 + i], b.row_nnz_ptr[r0 + i + 1]
                for j in range(z1 - z0):
                    f_idx.append((le, i, j))
                f_val.extend(b.col_global[z0:z1])
                v_val.extend(b.val[z0:z1])
            sl = slice(r0, r0 + n[e])
            y_val.extend(b.y[sl].astype(np.int64))
            o_val.extend(b.offset[sl])
            w_val.extend(b.weight[sl] if b.weight is not None else np.ones(n[e], np.float32))
            u_val.extend(b.uid[sl])
        d_idx = np.array(d_idx, np.int64).reshape(-1, 2)
        f_idx = np.array(f_idx, np.int64).reshape(-1, 3)
        ids = b.entity_ids[e0:e1]
        ent = np.array([int(s) for s in ids], np.int64) if int_entity_ids else np.array([s.encode() for s in ids], object)
        feats = {"entity": ent,
                 "uid": STV(d_idx, np.array(u_val, np.int64), None),
                 "offset": STV(d_idx, np.array(o_val, np.float32), None),
                 bag + "_indices": STV(f_idx, np.array(f_val, np.int64), None),
                 bag + "_values": STV(f_idx, np.array(v_val, np.float32), None)}
        if b.weight is not None:
            feats["weight"] = STV(d_idx, np.array(w_val, np.float32), None)
        labels = {"label": STV(d_idx, np.array(y_val, np.int64), None)}
        yield feats, labels


def run_reference(b: RawBatch, num_features, l2=1.0, regularize_bias=True, has_intercept=True, m=10,
                  max_iter=100, tol=1e-12, local=True, variance_mode=None, prior=None, batch_size=16,
                  int_entity_ids=False):
    """prepare_jobs -> TrainingJobConsumer -> fit, mirroring random_effect_lr_lbfgs_model.py:140-167."""
    lr = BinaryLogisticRegressionTrainer(regularize_bias=regularize_bias, lambda_l2=l2,
                                         precision=tol / np.finfo(float).eps,
                                         num_lbfgs_corrections=m, max_iter=max_iter,
                                         has_intercept=has_intercept)
    raw = []
    orig_fit = lr.fit

    def fit(**kw):
        res = orig_fit(**kw)
        raw.append((res, kw["theta_initial"].copy()))
        return res
    lr.fit = fit
    q = queue.Queue()
    consumer = jc.TrainingJobConsumer(lr, "fixture", q, enable_local_indexing=local,
                                      sparsity_threshold=1e-4, variance_mode=variance_mode)
    model_params = SimpleNamespace(partition_entity="entity", feature_bag="bag", offset_column_name="offset")
    schema = SimpleNamespace(uid_column_name="uid", label_column_name="label", weight_column_name="weight")
    out = []
    for jid in jc.prepare_jobs(lambda: tf_batches(b, batch_size, int_entity_ids=int_entity_ids),
                               model_params, schema, num_features, prior or {}, local, q, has_intercept):
        out.append(consumer(jid))
    return out, raw


def pack_results(b, out, raw, has_intercept, local):
    ic = 1 if has_intercept else 0
    theta, theta_thr, theta0, var, uniq, feat_ptr = [], [], [], [], [], [0]
    fval, nit, nfev, status, gmax = [], [], [], [], []
    for (eid, tr), ((res, variance), th0) in zip(out, raw):
        u = np.asarray(tr.unique_global_indices, np.int64)
        x = np.asarray(res[0], np.float64)
        if not local:  # global indexing: keep the entity's support only (job_consumers.py:87-99)
            sel = np.concatenate([[0], u + 1]) if ic else u
            x = x[sel]
            th0 = th0[sel]
        theta.append(x)
        theta0.append(th0)
        theta_thr.append(np.asarray(tr.theta, np.float64))
        if tr.variance is not None:
            var.append(np.asarray(tr.variance, np.float64))
        uniq.append(u)
        feat_ptr.append(feat_ptr[-1] + u.size)
        fval.append(res[1])
        nit.append(res[2]["nit"])
        nfev.append(res[2]["funcalls"])
        status.append(status_code(res[2]["task"]))
        gmax.append(np.max(np.abs(res[2]["grad"])))
    d = dict(theta=np.concatenate(theta), theta_thr=np.concatenate(theta_thr), theta0=np.concatenate(theta0),
             unique_global=np.concatenate(uniq), ent_feat_ptr=np.array(feat_ptr, np.int64),
             fval=np.array(fval), nit=np.array(nit, np.int32), nfev=np.array(nfev, np.int32),
             status=np.array(status, np.int32), gnorm=np.array(gmax),
             entity_ids=np.array([o[0] for o in out]))
    if var:
        d["variance"] = np.concatenate(var)
    return d


def save(name, b, opts, res, extra=None):
    d = dict(ent_row_ptr=b.ent_row_ptr, row_nnz_ptr=b.row_nnz_ptr, col_global=b.col_global, val=b.val,
             y=b.y, offset=b.offset, uid=b.uid, opts=np.array(json.dumps(opts)),
             scipy_version=np.array(scipy.__version__))
    if b.weight is not None:
        d["weight"] = b.weight
    d.update({"exp_" + k: v for k, v in res.items()})
    if extra:
        d.update(extra)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: E={b.E} N={b.N} Z={b.Z} nit mean {res['nit'].mean():.2f} max {res['nit'].max()} "
          f"nfev-nit-1 max {(res['nfev'] - res['nit'] - 1).max()} status {np.bincount(res['status'], minlength=5)} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def case(name, b, num_features, local=True, prior=None, int_ids=False, **kw):
    opts = dict(l2=1.0, regularize_bias=True, has_intercept=True, m=10, max_iter=100, tol=1e-12,
                variance_mode=None)
    opts.update(kw)
    out, raw = run_reference(b, num_features, local=local, prior=prior, int_entity_ids=int_ids, **opts)
    res = pack_results(b, out, raw, opts["has_intercept"], local)
    opts["num_features"] = num_features
    opts["local"] = local
    save(name, b, opts, res)
    return out


def reference_fixture_batch():
    """test/resources/grouped_per_member_train/data.tfrecord, decoded by hand."""
    return RawBatch(ent_row_ptr=[0, 2, 3], row_nnz_ptr=[0, 5, 7, 9],
                    col_global=[0, 7, 60, 80, 95, 34, 57, 10, 11],
                    val=[1, 2, 3, 5, 6.6, 1, 2, -3.5, 2.3], y=[0, 1, 1], offset=[0.5, 0.75, 0.2],
                    weight=[1, 2, 1], uid=[10, 20, 23], entity_ids=["100034", "100"])


def test_dataset(idx):
    """The in-test datasets of test_random_effect_lr_lbfgs_model.py:169-194 (values are data)."""
    if idx == 1:
        return RawBatch(ent_row_ptr=[0, 6], row_nnz_ptr=[0, 2, 4, 6, 8, 10, 12],
                        col_global=[0, 2, 0, 1, 1, 2, 0, 2, 1, 2, 0, 1],
                        val=[0.55, -0.95, 0.22, -1.05, 0.90, 0.50, 1.99, 0.48, 0.37, -1.64, 0.33, 0.17],
                        y=[1, 0, 1, 0, 0, 1], offset=[1.0, 2.0, 3.0, -1.0, 0.3, -0.7],
                        weight=[1.0, 0.8, 2.0, 3.0, 2.1, 1.7], uid=[1, 2, 3, 4, 5, 6], entity_ids=["xyz"])
    return RawBatch(ent_row_ptr=[0, 2, 4], row_nnz_ptr=[0, 3, 6, 8, 10],
                    col_global=[1, 5, 10, 1, 50, 99, 1, 3, 2, 20],
                    val=[0.3, -2.3, 0.9, 1.4, 99.8, -1.2, 1.23, 4.5, -1.0, 3.0],
                    y=[1, 0, 0, 0], offset=[1.0, 2.0, -1.0, -2.0], weight=[1.0, 0.8, 0.5, 0.74],
                    uid=[1234, 5678, 1345, 3214], entity_ids=["abc102", "zyz234"])


def main():
    # (a) the reference's own fixture, with the options its model test uses (l2=0.1, test :45)
    fx = reference_fixture_batch()
    case("ref_fixture_l2_0.1", fx, 100, l2=0.1, int_ids=True)
    case("ref_fixture_l2_0.1_global", fx, 100, local=False, l2=0.1, int_ids=True)
    case("ref_fixture_nobias_reg", fx, 100, l2=0.1, regularize_bias=False, int_ids=True)
    case("ref_fixture_maxiter1", fx, 100, l2=0.1, max_iter=1, int_ids=True)
    # (b) in-test datasets 1 and 2
    case("ref_dataset1", test_dataset(1), 3, l2=0.1)
    case("ref_dataset2", test_dataset(2), 100, l2=0.1)
    case("ref_dataset1_variance_full", test_dataset(1), 3, l2=0.0, regularize_bias=True, variance_mode="full")
    case("ref_dataset1_variance_simple", test_dataset(1), 3, l2=0.0, variance_mode="simple")
    # (c) seeded synthetic classes
    c2 = synthetic.make_batch(300, 16, 4, 1024, seed=synthetic.C2_SEED)
    case("c2_shipped_cfg", c2, 1024, l2=1.0, regularize_bias=False)            # lr-movieLens.yaml options
    case("c2_defaults", c2, 1024)                                              # REParams defaults
    case("c2_global_indexing", c2.select(np.arange(40)), 1024, local=False, regularize_bias=False)
    case("c2_l2_1e-3", c2.select(np.arange(150)), 1024, l2=1e-3, regularize_bias=False)
    case("c2_no_intercept", c2.select(np.arange(100)), 1024, has_intercept=False, regularize_bias=False)
    case("c2_maxiter1", c2.select(np.arange(100)), 1024, max_iter=1, regularize_bias=False)
    case("c2_maxiter3_m2", c2.select(np.arange(100)), 1024, max_iter=3, m=2, regularize_bias=False)
    case("c2_m3", c2.select(np.arange(150)), 1024, m=3, regularize_bias=False)
    c2w = synthetic.make_batch(150, 16, 4, 1024, seed=11, random_weights=True)
    case("c2_weights", c2w, 1024, regularize_bias=False)
    big = synthetic.make_batch(150, 16, 4, 1024, seed=12, l_offset=5.0, value_scale=3.0)
    case("c2_large_offsets", big, 1024, regularize_bias=False)
    c5 = synthetic.make_batch(60, 32, 8, 65536, seed=synthetic.C5_SEED)
    case("c5_mean_shape", c5, 65536, regularize_bias=False)
    tiny = synthetic.make_batch(300, 2, 2, 64, seed=13, size_dist="geometric")
    case("tiny_entities_regbias", tiny, 64, regularize_bias=True)
    case("tiny_entities_shipped_cfg", tiny, 64, regularize_bias=False)
    rag = synthetic.make_ragged_batch(150, seed=7)
    case("ragged", rag, 200, regularize_bias=False)
    case("ragged_variance_simple", rag.select(np.arange(40)), 200, variance_mode="simple")
    case("ragged_variance_full", rag.select(np.arange(40)), 200, variance_mode="full")
    mu = synthetic.make_movielens_like(60, "per_user", seed=100)
    case("ml_per_user", mu, 20, regularize_bias=False, int_ids=True)
    mm = synthetic.make_movielens_like(150, "per_movie", seed=101)
    case("ml_per_movie", mm, 24, regularize_bias=False, int_ids=True)
    zipf = synthetic.make_batch(40, 24, 8, 4096, seed=14, size_dist="zipf")
    case("zipf_tail", zipf, 4096, regularize_bias=False)

    # (d) warm start: train on the first half of each entity's features, then continue on the full data
    #     with the earlier model as prior (prepare_jobs:262-288); prior models keep extra features.
    ws = synthetic.make_batch(120, 16, 4, 1024, seed=15)
    out1 = case("warm_stage1", ws, 1024, regularize_bias=False, max_iter=3)
    prior = {eid: tr for eid, tr in out1}
    ws2 = synthetic.make_batch(120, 16, 4, 1024, seed=16)   # same ids, different samples/features
    opts = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=100, tol=1e-12,
                variance_mode=None)
    out2, raw2 = run_reference(ws2, 1024, local=True, prior=prior, **opts)
    res2 = pack_results(ws2, out2, raw2, True, True)
    opts.update(num_features=1024, local=True)
    pr_theta = np.concatenate([np.asarray(prior[e].theta, np.float64) for e in ws2.entity_ids])
    pr_idx = np.concatenate([np.asarray(prior[e].unique_global_indices, np.int64) for e in ws2.entity_ids])
    pr_ptr = np.concatenate([[0], np.cumsum([len(prior[e].unique_global_indices) for e in ws2.entity_ids])])
    save("warm_stage2", ws2, opts, res2,
         extra=dict(prior_theta=pr_theta, prior_unique_global=pr_idx, prior_feat_ptr=pr_ptr))

    # (e) scoring (InferenceJobConsumer, job_consumers.py:138-152) with the stage-2 models; entity
    #     ids 0..59 have a model, 60..119 do not (logit = offset).
    models = {eid: tr for eid, tr in out2[:60]}
    lr = BinaryLogisticRegressionTrainer(regularize_bias=True, lambda_l2=1.0, has_intercept=True)
    q = queue.Queue()
    schema_ns = SimpleNamespace(uid_column_name="uid", label_column_name="label", weight_column_name="weight",
                                prediction_score_column_name="predictionScore",
                                prediction_score_per_coordinate_column_name="predictionScorePerCoordinate")
    cons = jc.InferenceJobConsumer.__new__(jc.InferenceJobConsumer)
    cons.name, cons.num_features, cons.lr_model = "fixture", 1024, lr
    cons.schema_params, cons.job_count, cons.job_queue = schema_ns, 0, q
    model_params = SimpleNamespace(partition_entity="entity", feature_bag="bag", offset_column_name="offset")
    recs = []
    for jid in jc.prepare_jobs(lambda: tf_batches(ws2, 16), model_params, schema_ns, 1024, models, False, q, True):
        recs.extend(cons(jid))
    sc = np.array([r["predictionScore"] for r in recs], np.float64)
    pc = np.array([r["predictionScorePerCoordinate"] for r in recs], np.float64)
    np.savez_compressed(os.path.join(HERE, "score_stage2.npz"), exp_score=sc, exp_per_coord=pc,
                        exp_uid=np.array([r["uid"] for r in recs], np.int64), n_with_model=np.array(60))
    print("score_stage2:", sc.size, "records")


if __name__ == "__main__":
    main()
