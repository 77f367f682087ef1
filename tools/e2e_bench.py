"""End to end through the drop-in CLI on one MI355X: TFRecord partitions on disk -> python -m gdmix_amd.gdmix
--stage=random_effect --action=train -> photon-ml model Avro + score Avro, with the time of each phase.

    PYTHONPATH=. python tools/e2e_bench.py [entities] [partitions] [c2|zipf]

C2-shaped entities (n ~ Poisson(16), k = 4, D = 1024). The phases are timed by wrapping the model's own methods;
nothing is skipped: the active training data is read a second time for scoring, as the reference does.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

from gdmix_amd import gdmix as cli
from gdmix_amd import model as model_mod
from gdmix_amd import synthetic
from gdmix_amd.io.grouped_reader import write_grouped_partition

E = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 4
SHAPE = sys.argv[3] if len(sys.argv) > 3 else "c2"      # c2 | zipf (C5-shaped: Zipf sizes, k = 8, D = 65536)
DIM = 1024 if SHAPE == "c2" else 65536
md = {"features": [{"name": "bag", "dtype": "float", "shape": [DIM], "isSparse": True},
                   {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                   {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                   {"name": "ent", "dtype": "string", "shape": [], "isSparse": False}],
      "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}

phases = {}
events = []     # E2E_TIMELINE=1: (label, thread, start, end) of every wrapped call — who waits for whom


def timed(cls, name, label):
    import threading
    fn = getattr(cls, name)

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            t1 = time.perf_counter()
            phases[label] = phases.get(label, 0.0) + t1 - t
            events.append((label.strip(" ."), threading.current_thread().name, t, t1))
    setattr(cls, name, wrapper)


M = model_mod.RandomEffectLRLBFGSModel
timed(M, "_read", "read TFRecord (train + scoring passes)")
timed(M, "_solve_batch", "pack + solve + D2H")
timed(M, "_read_ahead", "  . decode ahead (+ upload)")
timed(M, "_train", "  . _train")
timed(M, "end_pipeline", "  . end_pipeline")
timed(M, "_save_model", "model Avro")
timed(M, "_predict", "scoring pass total (read + score + score Avro)")
timed(model_mod, "_write_scores", "score Avro")
timed(model_mod, "_model_coefficients_for_batch", "  . prior coefficients into the batch's index space")
timed(model_mod.ModelTable, "flatten", "  . model table -> flat arrays (in the model Avro time)")
timed(model_mod.ModelTable, "lookup", "  . id lookups")

from gdmix_amd import solver as solver_mod


def timed_sync(cls, name, label):
    import torch
    fn = getattr(cls, name)

    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            torch.cuda.synchronize()
            phases[label] = phases.get(label, 0.0) + time.perf_counter() - t
    setattr(cls, name, wrapper)


if os.environ.get("E2E_DETAIL"):     # attribute the device-side calls (adds synchronisation: totals get slightly worse)
    timed_sync(solver_mod.REDeviceSolver, "pack", "  . upload + pack")
    timed_sync(solver_mod.REDeviceSolver, "solve", "  . solve")
    timed_sync(solver_mod.REDeviceSolver, "score", "  . score")
    timed_sync(solver_mod.SolveResult, "to_host", "  . results D2H")

with tempfile.TemporaryDirectory() as d:
    t = time.perf_counter()
    b = synthetic.make_batch(E, 16, 4, 1024, seed=1) if SHAPE == "c2" else synthetic.make_batch(E, 32, 8, DIM, seed=synthetic.C5_SEED, size_dist="zipf")
    per = (E + parts - 1) // parts
    for k in range(parts):
        sub = b.select(np.arange(k * per, min(E, (k + 1) * per)))
        write_grouped_partition(os.path.join(d, "train", "active", f"partitionId={k}", "part-0.tfrecord"), sub, "ent", "bag",
                                weight_column_name=None)
    json.dump(md, open(os.path.join(d, "meta.json"), "w"))
    with open(os.path.join(d, "features.csv"), "w") as f:
        f.write("".join(f"f{i},\n" for i in range(DIM)))
    open(os.path.join(d, "plist.txt"), "w").write(",".join(str(k) for k in range(parts)))
    size = sum(os.path.getsize(os.path.join(r, x)) for r, _, fs in os.walk(os.path.join(d, "train")) for x in fs)
    print(f"{E} entities, {b.N} samples, {b.Z} nnz in {parts} partitions, {size / 1e6:.0f} MB of TFRecord "
          f"(generated + written in {time.perf_counter() - t:.1f} s)", flush=True)
    argv = ["gdmix", "--stage=random_effect", "--model_type=logistic_regression", "--uid_column_name=uid",
            "--label_column_name=response", "--prediction_score_column_name=predictionScore",
            f"--partition_list_file={d}/plist.txt", f"--training_data_dir={d}/train", f"--metadata_file={d}/meta.json",
            f"--output_model_dir={d}/models", "--feature_bag=bag", f"--feature_file={d}/features.csv",
            "--partition_entity=ent", "--regularize_bias=False", "--l2_reg_weight=1.0", f"--training_score_dir={d}/ts",
            "--action=train"]
    os.environ.pop("TF_CONFIG", None)
    import shutil
    timed(M, "_load_weights", "prior model Avro")
    for rep, label in enumerate(("cold start (pays for the HIP context and library load)", "cold start", "warm start from the model above")):
        if rep == 1:
            shutil.rmtree(os.path.join(d, "models"))
        phases.clear()
        events.clear()
        prof = None
        if os.environ.get("E2E_PROFILE") and rep >= 1:    # cProfile of the main thread (the pipeline's other threads are not seen)
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t = time.perf_counter()
        cli.run(argv)
        dt = time.perf_counter() - t
        if prof is not None:
            import io
            import pstats
            prof.disable()
            buf = io.StringIO()
            pstats.Stats(prof, stream=buf).sort_stats("cumulative").print_stats(int(os.environ["E2E_PROFILE"]))
            print("\n".join(l[:160] for l in buf.getvalue().splitlines() if l.strip()))
        out = sum(os.path.getsize(os.path.join(r, x)) for r, _, fs in os.walk(d) for x in fs if x.endswith(".avro"))
        print(f"{label}: {dt:.2f} s  {E / dt:,.0f} entities/s end to end ({out / 1e6:.0f} MB of Avro written)")
        for k, v in phases.items():
            print(f"    {k:52s} {v:7.2f} s")
        if os.environ.get("E2E_TIMELINE") and rep >= 1:
            for lab, th, a0, a1 in sorted(events, key=lambda e: e[2]):
                print(f"      {(a0 - t) * 1e3:8.1f} -> {(a1 - t) * 1e3:8.1f} ms ({(a1 - a0) * 1e3:7.1f})  {th:24s} {lab}")
