"""One pass of GDMix's coordinate descent on one node, through the drop-in CLI: global fixed effect -> per-user random effect
-> per-movie random effect (BASELINE config 3 "per-user + per-item random effects").

What gdmix-workflow schedules for gdmix-workflow/test/resources/lr-movieLens.yaml
(gdmix-workflow/src/gdmixworkflow/random_effect_workflow_generator.py:32-47,82-93; README.md:243-299):

    fixed_effect train            -> model, training scores, validation scores            (python -m gdmix_amd.gdmix --stage=fixed_effect)
    per-user partition (Spark)    -> offset := previous stage's predictionScore (FLOAT), joined on uid; group by user,
                                     Java-hash partitions, active/, passive/            (partitioner.py: OffsetUpdater.scala:105-129,
                                                                                          DataPartitioner.scala:203-380)
    per-user random_effect train  -> models, training scores (active [+ passive]), validation scores
    per-movie partition, train    -> the same with the per-user stage's scores as offsets

Every stage's predictionScore is the accumulated logit so far (X theta + offset), stored as Avro `float`; the next stage's offset is
that float. Only the first iteration is run, as the reference's workflow does: no per-coordinate score is subtracted
(OffsetUpdater's dFPerCoordinateScoreOpt = None), though update_offsets takes one.

The stages run through `gdmix_amd.gdmix` exactly as gdmix-workflow would start them (in this process, or as child processes);
the Spark partition job between them is partitioner.py. Nothing here touches the oracle: tests/chain_oracle.py restates the
chain on the CPU and tests/test_gpu_chain.py compares the files.
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

from . import partitioner
from .io import avro, tfrecord

USERS, MOVIES, RATINGS = 943, 1682, 100_000        # MovieLens-100K (scripts/download_process_movieLens_data.py:378-462)
D_MOVIE_FEATS = 20      # per-user bag: the movie's features — 19 genre flags + release_date / 2000 (:104,306-311,384)
D_USER_FEATS = 24       # per-movie bag: the user's features — age / 100, gender one-hot (2), occupation one-hot (21) (:341-346,386-387)
D_GLOBAL = D_MOVIE_FEATS + D_USER_FEATS
STAGES = ("global", "per_user", "per_movie")


def make_dataset(users=USERS, movies=MOVIES, ratings=RATINGS, seed=20240603, train_fraction=0.8, effect_scale=1.0):
    """MovieLens-100K-shaped synthetic ratings with planted effects on all three coordinates (real MovieLens cannot be fetched
    here): logit = g . [movie feats, user feats] + b + u_user . movie feats + a_user + v_movie . user feats + c_movie.
    User activity and movie popularity follow long tails (every user rates at least 20 titles, as in the data set).
    -> dict of flat per-sample arrays; bags are CSR over samples with ascending columns."""
    rng = np.random.default_rng(seed)
    # who rates what: user activity ~ lognormal with the data set's floor of 20, movie popularity ~ Zipf-like
    act = np.maximum(20, rng.lognormal(np.log(60.0), 0.9, users)) if ratings >= 20 * users else np.ones(users)
    user = rng.choice(users, ratings, p=act / act.sum())
    pop = 1.0 / (np.arange(movies) + 8.0) ** 0.9
    movie = rng.choice(movies, ratings, p=pop / pop.sum())
    # movie features: 1 - 3 genre flags of 19 (value 1) + release_date / 2000
    genres = np.zeros((movies, 19), bool)
    for mv in range(movies):
        genres[mv, rng.choice(19, rng.integers(1, 4), replace=False)] = True
    release = (rng.integers(1930, 1999, movies) / 2000.0).astype(np.float32)
    # user features: age / 100, gender one-hot, occupation one-hot
    age = (rng.integers(10, 70, users) / 100.0).astype(np.float32)
    gender = rng.integers(0, 2, users)
    occupation = rng.integers(0, 21, users)
    s = effect_scale
    g = rng.standard_normal(D_GLOBAL) * 0.5 * s
    b = -0.2
    u = rng.standard_normal((users, D_MOVIE_FEATS)) * 0.8 * s
    a = rng.standard_normal(users) * 0.7 * s
    v = rng.standard_normal((movies, D_USER_FEATS)) * 0.6 * s
    c = rng.standard_normal(movies) * 0.7 * s

    def movie_bag(mv):      # -> (columns, values) per sample over D_MOVIE_FEATS
        k = genres[mv].sum(1) + 1
        ptr = np.concatenate([[0], np.cumsum(k)]).astype(np.int64)
        cols = np.empty(ptr[-1], np.int64)
        vals = np.ones(ptr[-1], np.float32)
        rows, gcol = np.nonzero(genres[mv])
        slot = ptr[:-1][rows] + (np.arange(rows.size) - np.searchsorted(rows, rows))   # genre flags first, ascending
        cols[slot] = gcol
        cols[ptr[1:] - 1] = 19
        vals[ptr[1:] - 1] = release[mv]
        return ptr, cols, vals

    def user_bag(us):
        n = us.size
        ptr = np.arange(n + 1, dtype=np.int64) * 3
        cols = np.stack([np.zeros(n, np.int64), 1 + gender[us], 3 + occupation[us]], 1).ravel()
        vals = np.stack([age[us], np.ones(n, np.float32), np.ones(n, np.float32)], 1).ravel().astype(np.float32)
        return ptr, cols, vals

    mptr, mcols, mvals = movie_bag(movie)
    uptr, ucols, uvals = user_bag(user)
    # the global bag = movie features followed by user features (columns shifted by D_MOVIE_FEATS)
    km, ku = np.diff(mptr), np.diff(uptr)
    gptr = np.concatenate([[0], np.cumsum(km + ku)]).astype(np.int64)
    gcols = np.empty(gptr[-1], np.int64)
    gvals = np.empty(gptr[-1], np.float32)
    from .batch import _ranges
    gm = _ranges(gptr[:-1], km)
    gu = _ranges(gptr[:-1] + km, ku)
    gcols[gm], gvals[gm] = mcols, mvals
    gcols[gu], gvals[gu] = ucols + D_MOVIE_FEATS, uvals
    rows_m = np.repeat(np.arange(ratings), km)
    rows_u = np.repeat(np.arange(ratings), ku)
    rows_g = np.repeat(np.arange(ratings), km + ku)
    z = (np.bincount(rows_g, gvals * g[gcols], ratings) + b
         + np.bincount(rows_m, mvals * u[user[rows_m], mcols], ratings) + a[user]
         + np.bincount(rows_u, uvals * v[movie[rows_u], ucols], ratings) + c[movie])
    y = (rng.random(ratings) < 1.0 / (1.0 + np.exp(-z))).astype(np.int64)
    train = rng.random(ratings) < train_fraction
    return dict(n=ratings, uid=np.arange(ratings, dtype=np.int64) + 1000, user=user.astype(np.int64) + 1, movie=movie.astype(np.int64) + 1,
                response=y, train=train, true_logit=z,
                bags={"global": (gptr, gcols, gvals, D_GLOBAL), "per_user": (mptr, mcols, mvals, D_MOVIE_FEATS),
                      "per_movie": (uptr, ucols, uvals, D_USER_FEATS)},
                entity={"per_user": "user_id", "per_movie": "movie_id"})


def bag_rows(data, bag, rows):
    """The CSR bag restricted to `rows` (in that order)."""
    ptr, cols, vals, dim = data["bags"][bag]
    from .batch import _ranges
    k = np.diff(ptr)[rows]
    nz = _ranges(ptr[rows], k)
    return np.concatenate([[0], np.cumsum(k)]).astype(np.int64), cols[nz], vals[nz], dim


def _feature_file(path, dim, prefix):
    with open(path, "w") as f:
        f.write("".join(f"{prefix}{i},\n" for i in range(dim)))


def write_global_inputs(root, data, files=4):
    """The fixed-effect stage's inputs: per-record tf.train.Example files (train / validation), metadata, feature list."""
    d = os.path.join(root, "global")
    for name, rows in (("trainingData", np.flatnonzero(data["train"])), ("validationData", np.flatnonzero(~data["train"]))):
        os.makedirs(os.path.join(d, name), exist_ok=True)
        ptr, cols, vals, dim = bag_rows(data, "global", rows)
        cuts = np.linspace(0, rows.size, files + 1).astype(int)
        for f in range(files):
            recs = [tfrecord.encode_example({"uid": ("int64", [int(data["uid"][rows[i]])]), "response": ("int64", [int(data["response"][rows[i]])]),
                                             "global_indices": ("int64", cols[ptr[i]:ptr[i + 1]]), "global_values": ("float", vals[ptr[i]:ptr[i + 1]])})
                    for i in range(cuts[f], cuts[f + 1])]
            tfrecord.write_records(os.path.join(d, name, f"part-{f:05d}.tfrecord"), recs)
    md = {"features": [{"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                       {"name": "global", "dtype": "float", "shape": [D_GLOBAL], "isSparse": True}],
          "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}
    os.makedirs(os.path.join(d, "metadata"), exist_ok=True)
    with open(os.path.join(d, "metadata", "tensor_metadata.json"), "w") as f:
        json.dump(md, f)
    _feature_file(os.path.join(d, "featureList"), D_GLOBAL, "g")
    return d


def read_scores(score_dir):
    """Every score Avro file under score_dir -> (uid, predictionScore float32, predictionScorePerCoordinate float32, label)."""
    uid, sc, pc, lab = [], [], [], []
    for r, _, fs in sorted(os.walk(score_dir)):
        for fn in sorted(fs):
            if fn.endswith(".avro"):
                for rec in avro.read_file(os.path.join(r, fn)):
                    uid.append(rec["uid"])
                    sc.append(rec["predictionScore"])
                    pc.append(rec.get("predictionScorePerCoordinate", 0.0))
                    lab.append(rec.get("response"))
    return (np.array(uid, np.int64), np.array(sc, np.float32), np.array(pc, np.float32),
            np.array([np.nan if x is None else x for x in lab], np.float32))


def partition_stage(root, data, stage, prev_train_scores, prev_valid_scores, num_partitions=4, upper_bound=None):
    """The Spark partition job ahead of a random-effect stage (DataPartitioner.scala:203-380), host side: offsets from the previous
    stage's score files (update_offsets: inner join on uid, FLOAT), grouping by entity, Java-hash partition ids, active/ layout."""
    ent_col = data["entity"][stage]
    ent = data["user"] if stage == "per_user" else data["movie"]
    out = os.path.join(root, stage, "partition")
    parts = set()
    for name, mask, scores, split in (("trainingData", data["train"], prev_train_scores, True), ("validationData", ~data["train"], prev_valid_scores, False)):
        rows_all = np.flatnonzero(mask)
        s_uid, s_score, _, _ = read_scores(scores)
        keep, off = partitioner.update_offsets(data["uid"][rows_all], s_uid, s_score)
        rows = rows_all[keep]
        ptr, cols, vals, dim = bag_rows(data, stage, rows)
        # training data: an entity with more than upper_bound samples keeps group 0 (uid mod (count / upper_bound + 1)) as ACTIVE data,
        # trained on; its other groups are PASSIVE data, only scored (DataPartitioner.scala:322-380)
        batches = partitioner.build_batches(ent[rows], data["uid"][rows], data["response"][rows].astype(np.float32), off, None, ptr, cols, vals,
                                            num_partitions, upper_bound=upper_bound if split else None, split=split)
        partitioner.write_partitions(os.path.join(out, name), batches, ent_col, stage, int_entity_ids=True, weight_column_name=None)
        if split:
            parts |= {p for (_, p) in batches}
    dim = data["bags"][stage][3]
    md = {"features": [{"name": stage, "dtype": "float", "shape": [dim], "isSparse": True},
                       {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                       {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                       {"name": ent_col, "dtype": "long", "shape": [], "isSparse": False}],
          "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}
    os.makedirs(os.path.join(out, "metadata"), exist_ok=True)
    with open(os.path.join(out, "metadata", "tensor_metadata.json"), "w") as f:
        json.dump(md, f)
    with open(os.path.join(out, "partitionList.txt"), "w") as f:
        f.write(",".join(str(p) for p in sorted(parts)))
    _feature_file(os.path.join(out, "featureList"), dim, "m" if stage == "per_user" else "u")
    return out


COMMON = ["--model_type=logistic_regression", "--uid_column_name=uid", "--label_column_name=response",
          "--prediction_score_column_name=predictionScore", "--l2_reg_weight=1.0", "--regularize_bias=False",
          "--lbfgs_tolerance=1.0e-12", "--num_of_lbfgs_iterations=100", "--num_of_lbfgs_curvature_pairs=10"]


def stage_argv(root, stage):
    """The flat argv gdmix-workflow would hand the trainer for this stage of lr-movieLens.yaml (output under <root>/<stage>/)."""
    out = os.path.join(root, stage)
    if stage == "global":
        d = os.path.join(root, "global")
        return ["gdmix", "--stage=fixed_effect", "--action=train", f"--training_data_dir={d}/trainingData", f"--validation_data_dir={d}/validationData",
                f"--metadata_file={d}/metadata/tensor_metadata.json", f"--feature_file={d}/featureList", "--feature_bag=global",
                f"--output_model_dir={out}/models", f"--training_score_dir={out}/trainingScores", f"--validation_score_dir={out}/validationScores"] + COMMON
    p = os.path.join(out, "partition")
    return ["gdmix", "--stage=random_effect", "--action=train", f"--partition_list_file={p}/partitionList.txt", f"--training_data_dir={p}/trainingData",
            f"--validation_data_dir={p}/validationData", f"--metadata_file={p}/metadata/tensor_metadata.json", f"--feature_file={p}/featureList",
            f"--feature_bag={stage}", f"--partition_entity={'user_id' if stage == 'per_user' else 'movie_id'}",
            f"--output_model_dir={out}/models", f"--training_score_dir={out}/trainingScores", f"--validation_score_dir={out}/validationScores",
            "--enable_local_indexing=False", "--num_of_consumers=1", "--max_training_queue_size=10"] + COMMON


def run_stage(argv, child_process=False):
    """`python -m gdmix_amd.gdmix <argv>`: as a child process (how gdmix-workflow starts a stage, single_node/local_ops.py:42-54) or
    in this process."""
    os.environ.pop("TF_CONFIG", None)
    if child_process:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        cp = subprocess.run([sys.executable, "-m", "gdmix_amd.gdmix"] + argv[1:], cwd=root, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                            env=dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", "")))
        if cp.returncode != 0:
            raise RuntimeError(f"stage exited with {cp.returncode}: " + cp.stderr.decode(errors="replace")[-2000:])
    else:
        from . import gdmix as cli
        cli.run(argv)


def auc(label, score):
    """Area under the ROC curve, ties at half weight (what gdmix-data's AreaUnderROCCurveEvaluator reports per stage, README.md:295-299)."""
    label = np.asarray(label) > 0.5
    order = np.argsort(score, kind="stable")
    s = np.asarray(score)[order]
    ranks = np.empty(s.size, np.float64)
    i = 0
    while i < s.size:      # average ranks over ties
        j = i
        while j + 1 < s.size and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    pos = label[order]
    n1, n0 = int(pos.sum()), int((~pos).sum())
    return float((ranks[pos].sum() - n1 * (n1 + 1) / 2.0) / (n1 * n0)) if n1 and n0 else float("nan")


def run_chain(root, data, num_partitions=4, child_process=False, log=None, upper_bounds=None):
    """global -> per_user -> per_movie under `root`; -> {stage: {"s", "partition_s", "train_auc", "validation_auc"}, "total_s"}.
    upper_bounds: {stage: active-data bound per entity} (the rest of a larger entity's samples is passive data)."""
    os.makedirs(root, exist_ok=True)
    write_global_inputs(root, data)
    out = {}
    prev = None
    t_all = time.perf_counter()
    for stage in STAGES:
        t_part = 0.0
        if stage != "global":
            t = time.perf_counter()
            partition_stage(root, data, stage, os.path.join(root, prev, "trainingScores"), os.path.join(root, prev, "validationScores"), num_partitions,
                            upper_bound=(upper_bounds or {}).get(stage))
            t_part = time.perf_counter() - t
        t = time.perf_counter()
        run_stage(stage_argv(root, stage), child_process)
        dt = time.perf_counter() - t
        r = {"s": dt, "partition_s": t_part}
        for which, d in (("train", "trainingScores"), ("validation", "validationScores")):
            uid, sc, _, lab = read_scores(os.path.join(root, stage, d))
            r[which + "_auc"] = auc(lab, sc)
            r[which + "_samples"] = int(uid.size)
        out[stage] = r
        if log:
            log(f"{stage}: {dt:.2f} s (+ {t_part:.2f} s partition), AUC train {r['train_auc']:.4f} validation {r['validation_auc']:.4f}")
        prev = stage
    out["total_s"] = time.perf_counter() - t_all
    return out
