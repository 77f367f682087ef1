"""Decode throughput of the entity-grouped TFRecord readers on a C2-shaped partition (host only).

    PYTHONPATH=. python tools/io_bench.py [entities] [threads ...]
"""
import os
import sys
import tempfile
import time

from gdmix_amd import synthetic

from gdmix_amd.io.grouped_reader import read_grouped_partition, write_grouped_partition

E = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
threads = [int(x) for x in sys.argv[2:]] or [1, 4, 0]
md = {"features": [{"name": "bag", "dtype": "float", "shape": [1024], "isSparse": True},
                   {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                   {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                   {"name": "ent", "dtype": "string", "shape": [], "isSparse": False}],
      "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}
b = synthetic.make_batch(E, 16, 4, 1024, seed=1)
with tempfile.TemporaryDirectory() as d:
    t = time.perf_counter()
    nfiles = 8
    per = (E + nfiles - 1) // nfiles
    import numpy as np
    for i in range(nfiles):
        part = b.select(np.arange(i * per, min(E, (i + 1) * per)))
        write_grouped_partition(os.path.join(d, f"part-{i:05d}.tfrecord"), part, "ent", "bag", weight_column_name=None)
    size = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
    print(f"{E} entities, {b.N} samples, {b.Z} nnz, {size / 1e6:.1f} MB in {nfiles} files (written in {time.perf_counter() - t:.1f} s)")
    args = (d, md, "ent", "bag", "offset", "uid", "response", None)
    if E <= 20000:
        t = time.perf_counter()
        read_grouped_partition(*args, native=False)
        dt = time.perf_counter() - t
        print(f"python reader            {dt:8.3f} s  {size / dt / 1e6:9.1f} MB/s  {E / dt:12.0f} entities/s")
    for th in threads:
        best = None
        for _ in range(3):
            t = time.perf_counter()
            r = read_grouped_partition(*args, native=True, threads=th, check_crc=True)
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        print(f"native, {th or 'all':>3} threads, crc {best:8.3f} s  {size / best / 1e6:9.1f} MB/s  {E / best:12.0f} entities/s")

# ---- Avro writers: model file (C2-shaped coefficients) and score file ------------------------------------------
from gdmix_amd.model import ModelTable, _export_models_to_avro, _write_scores
from gdmix_amd.io import avro
from types import SimpleNamespace
import numpy as np
rng = np.random.default_rng(0)
d = rng.integers(40, 64, E)
feat_ptr = np.concatenate([[0], np.cumsum(d)])
coef_ptr = feat_ptr + np.arange(E + 1)
idx = rng.integers(0, 1024, feat_ptr[-1])
theta = rng.standard_normal(coef_ptr[-1])
table = ModelTable()
table.add_chunk([str(i) for i in range(E)], theta, coef_ptr, idx, feat_ptr)
fl = [(f"feature{j}", "") for j in range(1024)]
sp = SimpleNamespace(uid_column_name="uid", prediction_score_column_name="predictionScore", label_column_name="response",
                     weight_column_name="weight", prediction_score_per_coordinate_column_name="predictionScorePerCoordinate")
schema = avro.inference_output_schema(sp, has_weight=False)
n = b.N
with tempfile.TemporaryDirectory() as d2:
    for native in ((False, True) if E <= 20000 else (True,)):
        t = time.perf_counter()
        _export_models_to_avro(os.path.join(d2, "m.avro"), table, fl, True, False, native=native)
        dt = time.perf_counter() - t
        sz = os.path.getsize(os.path.join(d2, "m.avro"))
        print(f"model export  {'native' if native else 'python'}  {dt:8.3f} s  {sz / dt / 1e6:9.1f} MB/s  {E / dt:12.0f} entities/s ({sz / 1e6:.0f} MB)")
        t = time.perf_counter()
        _write_scores(os.path.join(d2, "s.avro"), schema, sp, b.uid, b.offset, b.y, None, b.offset, native=native)
        dt = time.perf_counter() - t
        print(f"score export  {'native' if native else 'python'}  {dt:8.3f} s  {n / dt:12.0f} samples/s")
