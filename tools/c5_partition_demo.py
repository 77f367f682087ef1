"""C5-shaped data from a flat sample table to trained models, without Spark: synthetic Zipf-sized entities as one row per
sample -> gdmix_amd.partitioner (DataPartitioner's grouping, active / passive bounding, Java-hash partition id, the
`active/partitionId=K/` layout) -> python -m gdmix_amd.gdmix --stage=random_effect --action=train on one MI355X.

    PYTHONPATH=. python tools/c5_partition_demo.py [entities] [partitions] [upper_bound]
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

from gdmix_amd import gdmix as cli
from gdmix_amd import partitioner, synthetic
from gdmix_amd.io import avro

E = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 16
upper = int(sys.argv[3]) if len(sys.argv) > 3 else 0     # > 0: entities with more samples are bounded (rest goes to passive)
D = 65536
t0 = time.perf_counter()
b = synthetic.make_batch(E, 32, 8, D, seed=synthetic.C5_SEED, size_dist="zipf", with_uid=True)
n = b.ent_n()
entity = np.repeat(np.arange(E, dtype=np.int64) * 7919 + 13, n)    # one row per sample, as the fixed-effect stage leaves them
perm = np.random.default_rng(1).permutation(b.N)                   # and in no particular order
inv_rows = perm
rnp = b.row_nnz_ptr
k = np.diff(rnp)[inv_rows]
from gdmix_amd.batch import _ranges  # noqa: E402
nz = _ranges(rnp[inv_rows], k)
flat = dict(entity=entity[inv_rows], uid=b.uid[inv_rows], label=b.y[inv_rows], offset=b.offset[inv_rows], weight=None,
            row_nnz_ptr=np.concatenate([[0], np.cumsum(k)]), col_global=b.col_global[nz], val=b.val[nz])
print(f"flat table: {b.N} samples of {E} entities, {b.Z} non-zeros ({time.perf_counter() - t0:.1f} s to generate)", flush=True)
y1 = np.add.reduceat(b.y.astype(np.float64), b.ent_row_ptr[:-1])
print(f"parity classes: W (both labels present) {int(((y1 > 0) & (y1 < n)).sum())}, D (all labels equal) {int(((y1 == 0) | (y1 == n)).sum())}; "
      f"largest entity {int(n.max())} samples / {int(b.ent_nnz().max())} non-zeros")

with tempfile.TemporaryDirectory() as d:
    t = time.perf_counter()
    batches = partitioner.build_batches(num_partitions=parts, upper_bound=upper or None, **flat)
    t_build = time.perf_counter() - t
    t = time.perf_counter()
    paths = partitioner.write_partitions(os.path.join(d, "train"), batches, "ent", "bag", int_entity_ids=True, weight_column_name=None)
    t_write = time.perf_counter() - t
    size = sum(os.path.getsize(p) for p in paths)
    act = {p: bb.E for (s, p), bb in batches.items() if s == partitioner.ACTIVE}
    pas = {p: bb.E for (s, p), bb in batches.items() if s == partitioner.PASSIVE}
    work = np.array([batches[(partitioner.ACTIVE, p)].Z for p in sorted(act)])
    print(f"partitioner: grouped + hashed in {t_build:.1f} s, {len(paths)} files / {size / 1e6:.0f} MB written in {t_write:.1f} s; "
          f"active records per partition {min(act.values())}..{max(act.values())}, passive records {sum(pas.values())}; "
          f"non-zeros per partition {work.min()}..{work.max()} (max / mean {work.max() / work.mean():.2f})", flush=True)
    md = {"features": [{"name": "bag", "dtype": "float", "shape": [D], "isSparse": True},
                       {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                       {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                       {"name": "ent", "dtype": "long", "shape": [], "isSparse": False}],
          "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}
    json.dump(md, open(os.path.join(d, "meta.json"), "w"))
    with open(os.path.join(d, "features.csv"), "w") as f:
        f.write("".join(f"f{i},\n" for i in range(D)))
    open(os.path.join(d, "plist.txt"), "w").write(",".join(str(p) for p in sorted(act)))
    argv = ["gdmix", "--stage=random_effect", "--model_type=logistic_regression", "--uid_column_name=uid", "--label_column_name=response",
            "--prediction_score_column_name=predictionScore", f"--partition_list_file={d}/plist.txt", f"--training_data_dir={d}/train",
            f"--metadata_file={d}/meta.json", f"--output_model_dir={d}/models", "--feature_bag=bag", f"--feature_file={d}/features.csv",
            "--partition_entity=ent", "--regularize_bias=False", "--l2_reg_weight=1.0", f"--training_score_dir={d}/ts", "--action=train"]
    os.environ.pop("TF_CONFIG", None)
    for rep in range(2):
        t = time.perf_counter()
        cli.run(argv)
        dt = time.perf_counter() - t
        if rep == 0:
            import shutil
            shutil.rmtree(os.path.join(d, "models"))
    n_models = sum(1 for p in sorted(act) for _ in avro.read_file(os.path.join(d, "models", f"part-{p:05d}.avro"))) if E <= 20000 else sum(act.values())
    print(f"train (cold, second pass): {dt:.2f} s = {sum(act.values()) / dt:,.0f} entities/s end to end over {len(act)} partitions, {n_models} models written")
