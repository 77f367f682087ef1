"""Materialise a host RawBatch as the partition directory the trainer reads (contract B3):
<dir>/train/active/partitionId=K/part-00000.tfrecord, metadata JSON, feature list CSV, partition list — entities assigned to
partitions by the Java hash of their id (PartitionUtils.scala:31-37), as gdmix-data's DataPartitioner leaves them. Used by
bench.py's end-to-end legs, tools/ and tests; nothing of the solve."""
import json
import os

import numpy as np

from .io.grouped_reader import write_grouped_partition
from .partitioner import partition_ids


def write_partition_dir(root, batch, num_partitions, dim, int_entity_ids=True):
    """-> (argv for gdmix_amd.gdmix --action=train, {partition id: entity indices of `batch` in file order}, bytes written)."""
    ids = np.asarray([int(x) for x in batch.entity_ids], np.int64) if int_entity_ids else list(batch.entity_ids)
    pid = partition_ids(ids, num_partitions)
    members, size = {}, 0
    for k in range(num_partitions):
        own = np.flatnonzero(pid == k)
        if own.size == 0:
            continue
        members[k] = own
        path = os.path.join(root, "train", "active", f"partitionId={k}", "part-00000.tfrecord")
        write_grouped_partition(path, batch.select(own), "ent", "bag", weight_column_name=None, int_entity_ids=int_entity_ids)
        size += os.path.getsize(path)
    md = {"features": [{"name": "bag", "dtype": "float", "shape": [int(dim)], "isSparse": True},
                       {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                       {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                       {"name": "ent", "dtype": "long" if int_entity_ids else "string", "shape": [], "isSparse": False}],
          "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}
    with open(os.path.join(root, "meta.json"), "w") as f:
        json.dump(md, f)
    with open(os.path.join(root, "features.csv"), "w") as f:
        f.write("".join(f"f{i},\n" for i in range(int(dim))))
    with open(os.path.join(root, "plist.txt"), "w") as f:
        f.write(",".join(str(k) for k in sorted(members)))
    argv = ["gdmix", "--stage=random_effect", "--model_type=logistic_regression", "--uid_column_name=uid", "--label_column_name=response",
            "--prediction_score_column_name=predictionScore", f"--partition_list_file={root}/plist.txt", f"--training_data_dir={root}/train",
            f"--metadata_file={root}/meta.json", f"--output_model_dir={root}/models", "--feature_bag=bag", f"--feature_file={root}/features.csv",
            "--partition_entity=ent", "--regularize_bias=False", "--l2_reg_weight=1.0", f"--training_score_dir={root}/ts", "--action=train"]
    return argv, members, size
