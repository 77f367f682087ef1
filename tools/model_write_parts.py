"""Where the time of one model file goes (C2-shaped partition of 125 k entities, 64 coefficients each): the Python side
(ModelTable.flatten, id bytes) and the native writer at several thread counts.   PYTHONPATH=. python tools/model_write_parts.py"""
import os
import sys
import tempfile
import time

import numpy as np

from gdmix_amd import model as M
from gdmix_amd.io import avro, native_reader

E = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
p = 64
rng = np.random.default_rng(0)
ids = [str(1_000_000 + i) for i in range(E)]
coef_ptr = np.arange(E + 1, dtype=np.int64) * p
theta = rng.standard_normal(E * p)
feat_ptr = np.arange(E + 1, dtype=np.int64) * (p - 1)
idx = np.sort(rng.integers(0, 1024, (E, p - 1)), axis=1).ravel().astype(np.int64)
table = M.ModelTable()
t = time.perf_counter(); table.add_chunk(ids, theta, coef_ptr, idx, feat_ptr, None); print(f"add_chunk          {1e3 * (time.perf_counter() - t):7.1f} ms")
feature_list = [(f"f{i}", "") for i in range(1024)]
with tempfile.TemporaryDirectory() as d:
    for rep in range(2):
        t = time.perf_counter(); flat = table.flatten(); t_flat = time.perf_counter() - t
        ids2, coef_beg, coef_cnt, var_beg, feat_beg, mean, variance, fidx = flat
        t = time.perf_counter(); native_reader._ids_to_bytes(ids2); t_ids = time.perf_counter() - t
        print(f"flatten            {1e3 * t_flat:7.1f} ms\nids -> bytes       {1e3 * t_ids:7.1f} ms")
        prefix = [avro.enc_string(n) + avro.enc_string(tm) for (n, tm) in feature_list]
        header, sync = avro.container_header(avro.BAYESIAN_LINEAR_MODEL_SCHEMA, "null", None)
        icpt = avro.enc_string("(INTERCEPT)") + avro.enc_string("")
        head_class = avro.enc_long(1) + avro.enc_string("x")
        loss = avro.enc_long(1) + avro.enc_string("")
        for th in (1, 8, 32, 64, 0):
            path = os.path.join(d, f"m{th}.avro")
            t = time.perf_counter()
            native_reader.write_models_avro(path, header, sync, ids2, coef_beg, coef_cnt, mean, feat_beg, fidx, prefix, icpt, head_class, loss,
                                            True, 1e-4, threads=th)
            dt = time.perf_counter() - t
            print(f"write_models_avro threads={th:3d} {1e3 * dt:7.1f} ms  {os.path.getsize(path) / dt / 1e9:5.2f} GB/s ({os.path.getsize(path) / 1e6:.0f} MB)")
        for k in range(2):
            t = time.perf_counter()
            m = native_reader.read_models_avro(path, len(header), sync, False, prefix, icpt, True)
            dt = time.perf_counter() - t
            print(f"read_models_avro of that file                    {1e3 * dt:7.1f} ms  {E / dt / 1e6:5.2f} M entities/s")
            del m
        data = open(path, "rb").read()
        for k in range(2):
            t = time.perf_counter()
            with open(os.path.join(d, "raw.bin"), "wb") as fh:
                fh.write(data)
            dt = time.perf_counter() - t
            print(f"plain write() of the same {len(data) / 1e6:.0f} MB            {1e3 * dt:7.1f} ms  {len(data) / dt / 1e9:5.2f} GB/s")
        t = time.perf_counter()
        M._export_models_to_avro(os.path.join(d, "full.avro"), table, feature_list, True, False)
        print(f"_export_models_to_avro (all of it) {1e3 * (time.perf_counter() - t):7.1f} ms")
