"""End to end through the CLI for the fixed-effect stage on one MI355X: per-record TFRecord files on disk ->
python -m gdmix_amd.gdmix --stage=fixed_effect --action=train -> model Avro + score Avro, with the time of each phase.

    PYTHONPATH=. python tools/e2e_fe_bench.py [samples] [nnz_per_sample] [features]
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

from gdmix_amd import fe_model as fm
from gdmix_amd import fixed_effect as fe
from gdmix_amd import gdmix as cli
from gdmix_amd.io import tfrecord

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
D = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
phases = {}


def timed(cls, name, label):
    fn = getattr(cls, name)

    def wrapper(*a, **kw):
        t = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            phases[label] = phases.get(label, 0.0) + time.perf_counter() - t
    setattr(cls, name, wrapper)


timed(fm.FixedEffectLRModelLBFGS, "_read", "read per-record TFRecord (train + validation)")
timed(fe.FixedEffectDeviceSolver, "fit_stepping", "upload + pack + L-BFGS")
timed(fm.FixedEffectLRModelLBFGS, "_score_and_write", "score + score Avro (train + validation)")
timed(fm.FixedEffectLRModelLBFGS, "_save_model", "model Avro")
timed(fm.FixedEffectLRModelLBFGS, "_load_model", "prior model")

rng = np.random.default_rng(0)
cols = rng.integers(0, D, (n, k))
vals = rng.standard_normal((n, k)).astype(np.float32)
w = rng.standard_normal(D) * 0.2
y = (rng.random(n) < 1 / (1 + np.exp(-(vals * w[cols]).sum(1)))).astype(np.int64)
off = (0.1 * rng.standard_normal(n)).astype(np.float32)
with tempfile.TemporaryDirectory() as d:
    t = time.perf_counter()
    files = 4
    cuts = np.linspace(0, n, files + 1).astype(int)
    for name in ("train", "valid"):
        os.makedirs(os.path.join(d, name))
        for f in range(files):
            recs = [tfrecord.encode_example({"uid": ("int64", [i]), "offset": ("float", [float(off[i])]), "response": ("int64", [int(y[i])]),
                                             "global_indices": ("int64", cols[i]), "global_values": ("float", vals[i])})
                    for i in range(cuts[f], cuts[f + 1] if name == "train" else cuts[f] + (cuts[f + 1] - cuts[f]) // 4)]
            tfrecord.write_records(os.path.join(d, name, f"part-{f:05d}.tfrecord"), recs)
    size = sum(os.path.getsize(os.path.join(r, x)) for r, _, fs in os.walk(d) for x in fs)
    print(f"{n} training samples x {k} non-zeros, {D} features, {size / 1e6:.0f} MB of TFRecord written in {time.perf_counter() - t:.1f} s", flush=True)
    md = {"features": [{"name": "uid", "dtype": "long", "shape": [], "isSparse": False}, {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                       {"name": "global", "dtype": "float", "shape": [D], "isSparse": True}],
          "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}
    json.dump(md, open(os.path.join(d, "meta.json"), "w"))
    with open(os.path.join(d, "features.csv"), "w") as f:
        f.write("".join(f"f{i},\n" for i in range(D)))
    argv = ["gdmix", "--stage=fixed_effect", "--action=train", "--model_type=logistic_regression", "--uid_column_name=uid",
            "--label_column_name=response", "--prediction_score_column_name=predictionScore", f"--training_data_dir={d}/train",
            f"--validation_data_dir={d}/valid", f"--metadata_file={d}/meta.json", f"--output_model_dir={d}/model", "--feature_bag=global",
            f"--feature_file={d}/features.csv", f"--training_score_dir={d}/ts", f"--validation_score_dir={d}/vs", "--l2_reg_weight=1.0",
            "--num_of_lbfgs_iterations=30"]
    os.environ.pop("TF_CONFIG", None)
    import shutil
    for rep, label in enumerate(("first pass (HIP context, library load)", "cold", "warm start from the saved model")):
        if rep == 1:
            shutil.rmtree(os.path.join(d, "model"))
        phases.clear()
        t = time.perf_counter()
        cli.run(argv)
        dt = time.perf_counter() - t
        print(f"{label}: {dt:.2f} s  ({n / dt:,.0f} training samples/s end to end)")
        for kk, v in phases.items():
            print(f"    {kk:52s} {v:7.2f} s")
