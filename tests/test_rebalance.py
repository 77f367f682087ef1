"""Entity re-balancing (gdmix_amd/rebalance.py): the plan on its own, and the whole exchange -> solve -> give back
round trip on two gloo ranks, whose model files must equal those of the run without re-balancing. CPU only."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from gdmix_amd import synthetic
from gdmix_amd.io import avro
from gdmix_amd.io.grouped_reader import write_grouped_partition
from gdmix_amd.rebalance import choose_entities, plan_transfers, wire_tensors, wire_to_raw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_moves_surplus_to_deficit_and_leaves_balanced_ranks_alone():
    T = plan_transfers([100.0, 20.0, 60.0, 60.0])
    assert T.shape == (4, 4) and np.allclose(T.sum(), 40.0)
    assert np.allclose(T[0], [0, 40, 0, 0])                     # the only donor gives to the only rank in deficit
    assert not T[2].any() and not T[3].any() and not T[:, 0].any()
    assert not plan_transfers([50, 51, 49, 50]).any()           # within tolerance: nothing moves
    T = plan_transfers([90, 90, 10, 10])
    assert np.allclose(T.sum(axis=1), [40, 40, 0, 0]) and np.allclose(T.sum(axis=0), [0, 0, 40, 40])
    assert not plan_transfers([]).any() and not plan_transfers([7.0]).any()


def test_choose_entities_deals_out_the_small_ones_and_keeps_the_giants():
    cost = np.array([1000, 3, 5, 2, 8, 400, 1, 6], np.float64)
    to = choose_entities(cost, [0.0, 10.0, 9.0])
    assert to[0].size == 0
    assert cost[to[1]].sum() <= 10.0 and cost[to[2]].sum() <= 9.0
    moved = np.concatenate(to)
    assert len(set(moved.tolist())) == moved.size and 0 not in moved and 5 not in moved
    assert sorted(cost[to[1]].tolist()) == [1, 2, 3]            # ascending cost: 1 + 2 + 3 (the next, 5, would exceed 10)
    assert sorted(cost[to[2]].tolist()) == [5]                  # continues where the previous destination stopped


def test_wire_form_round_trip():
    """What travels: the 32-bit wire form of a partition (no ids, no sample ids); back to a host batch it is the same numbers."""
    import torch
    b = synthetic.make_ragged_batch(40, seed=2)
    w = wire_tensors(b, torch.device("cpu"))
    assert w["E"] == b.E and w["row_nnz"].dtype == torch.int32 and w["col_global"].dtype == torch.int32 and w["y"].dtype == torch.float32
    r = wire_to_raw(w)
    for k in ("ent_row_ptr", "row_nnz_ptr", "col_global", "val", "y", "offset", "weight"):
        np.testing.assert_array_equal(getattr(r, k), getattr(b, k))
    assert r.uid is None and r.entity_ids is None


def _run(tmp_path, tag, rebalance, extra=(), device_solver=False):
    out = tmp_path / tag
    argv = json.load(open(tmp_path / "argv.json"))
    argv = [a for a in argv if not a.startswith("--output_model_dir")] + [f"--output_model_dir={out / 'models'}",
                                                                         f"--rebalance_entities={rebalance}"] + list(extra)
    os.makedirs(out, exist_ok=True)
    json.dump(argv, open(out / "argv.json", "w"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TF_CONFIG", None)
    if device_solver:
        env["GDMIX_TEST_DEVICE_SOLVER"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29613", os.path.join(ROOT, "tests", "_dist_worker.py"), str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    return out


def _skewed_partitions(tmp_path):
    # partition 0 (rank 0) is ~10x heavier than partition 1 (rank 1); partition 2 exists only for rank 0, so rank 1
    # idles through the second round and still receives work
    heavy = synthetic.make_batch(300, 24, 4, 256, seed=1)
    light = synthetic.make_batch(30, 8, 4, 256, seed=2, entity_id_base=10_000)
    third = synthetic.make_batch(120, 16, 4, 256, seed=3, entity_id_base=20_000)
    md = {"features": [{"name": "bag", "dtype": "float", "shape": [256], "isSparse": True},
                       {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                       {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                       {"name": "ent", "dtype": "string", "shape": [], "isSparse": False}],
          "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}
    json.dump(md, open(tmp_path / "meta.json", "w"))
    with open(tmp_path / "features.csv", "w") as f:
        f.write("".join(f"f{i},\n" for i in range(256)))
    for k, b in enumerate((heavy, light, third)):
        write_grouped_partition(str(tmp_path / "train" / "active" / f"partitionId={k}" / "part-0.tfrecord"), b, "ent", "bag",
                                weight_column_name=None)
    open(tmp_path / "partitionList.txt", "w").write("0,1,2")
    argv = ["gdmix", "--stage=random_effect", "--action=train", "--uid_column_name=uid", "--label_column_name=response",
            f"--partition_list_file={tmp_path / 'partitionList.txt'}", f"--training_data_dir={tmp_path / 'train'}",
            f"--metadata_file={tmp_path / 'meta.json'}", "--feature_bag=bag", f"--feature_file={tmp_path / 'features.csv'}",
            "--partition_entity=ent", "--regularize_bias=False", "--disable_random_effect_scoring_after_training=True",
            f"--training_score_dir={tmp_path / 'ts'}", "--prediction_score_column_name=predictionScore", "--output_model_dir=x"]
    json.dump(argv, open(tmp_path / "argv.json", "w"))
    return heavy, light, third


def test_two_ranks_rebalance_skewed_partitions_and_write_the_same_models(tmp_path):
    heavy, light, third = _skewed_partitions(tmp_path)
    plain = _run(tmp_path, "plain", False)
    moved = _run(tmp_path, "rebalanced", True)
    for k, b in enumerate((heavy, light, third)):
        a = list(avro.read_file(str(plain / "models" / f"part-{k:05d}.avro")))
        r = list(avro.read_file(str(moved / "models" / f"part-{k:05d}.avro")))
        assert len(a) == b.E
        assert a == r                                           # same entities, same order, bit-identical coefficients
    res = json.load(open(moved / "result.json"))
    assert res["per_rank"] == [[0, 2], [1]]
    rounds = res["rebalance"]
    # round 1: rank 0 gives away entities of the heavy partition; round 2: rank 1 has no partition and takes half
    assert rounds[0][0]["sent"][1] > 0 and rounds[1][0]["received"][0] > 0
    assert rounds[0][1]["sent"][1] > 0 and rounds[1][1]["entities"] == 0 and rounds[1][1]["received"][0] > 0
    assert not rounds[0][0]["with_prior"]
    # second pass over the same directories with another regularisation weight: a warm start from the models above. The
    # prior models of the travelling entities travel with them, so the result is again bit-identical to the plain run.
    plain = _run(tmp_path, "plain", False, ["--l2_reg_weight=3.0"])
    moved = _run(tmp_path, "rebalanced", True, ["--l2_reg_weight=3.0"])
    for k, b in enumerate((heavy, light, third)):
        a = list(avro.read_file(str(plain / "models" / f"part-{k:05d}.avro")))
        r = list(avro.read_file(str(moved / "models" / f"part-{k:05d}.avro")))
        assert len(a) == b.E and a == r
    rounds = json.load(open(moved / "result.json"))["rebalance"]
    assert rounds[0][0]["with_prior"] and rounds[0][0]["sent"][1] > 0
    r1 = rounds[1][0]    # rank 1, first round: its own light partition plus what rank 0 gave away
    assert r1["prior_models"] == r1["solved"] > r1["entities"]     # every entity came with its model


@pytest.mark.gpu
def test_rccl_branch_of_the_exchange_runs_on_device_memory():
    """The `nccl` (= RCCL) branch of the exchange has only one GPU to run on here: a single-rank process group, the wire form of a
    partition in HBM, every array travelling rank 0 -> rank 0 through RCCL's all_to_all_single on device tensors, widen + pack +
    solve of what came back, give_back on device tensors — no device-to-host copy of entity payload in between
    (tests/_nccl_worker.py)."""
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TF_CONFIG", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_nccl_worker.py")], env=env, timeout=600, cwd=ROOT,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "nccl single-rank exchange ok" in r.stdout


@pytest.mark.gpu
def test_two_ranks_rebalance_over_rccl_when_the_box_has_two_gpus(tmp_path):
    """The same two-rank run on the product path: one process per GPU, the device solver, the exchange over RCCL on device tensors.
    Needs two devices (the driver's 8-GPU box); on the 1-GPU box the single-rank RCCL test above is what can run."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: one process per GPU")
    _two_ranks_on_the_product_path(tmp_path, "nccl")


@pytest.mark.gpu
def test_two_ranks_rebalance_on_the_product_path_sharing_one_gpu(tmp_path):
    """The product path of the exchange with the device solver and two ranks on a 1-GPU box: both ranks on cuda:0
    (GDMIX_RANKS_SHARE_DEVICE, a test hook), the process group gloo, the exchange's device tensors staged through the host by
    rebalance._Comm — everything else (wire form in HBM, gather of the travelling entities on the device, widen + pack + solve where
    they land, the measured cost model from the second round on, give-back, one read-back by the owner) is what RCCL ranks run."""
    _two_ranks_on_the_product_path(tmp_path, "gloo")


def _two_ranks_on_the_product_path(tmp_path, backend):
    heavy, light, third = _skewed_partitions(tmp_path)
    import os as _os
    share = {"GDMIX_RANKS_SHARE_DEVICE": "1"} if backend == "gloo" else {}
    old = {k: _os.environ.get(k) for k in share}
    _os.environ.update(share)
    try:
        plain = _run(tmp_path, "plain", False, device_solver=True)
        moved = _run(tmp_path, "rebalanced", True, device_solver=True)
    finally:
        for k, v in old.items():
            if v is None:
                _os.environ.pop(k, None)
            else:
                _os.environ[k] = v
    for k, b in enumerate((heavy, light, third)):
        a = list(avro.read_file(str(plain / "models" / f"part-{k:05d}.avro")))
        r = list(avro.read_file(str(moved / "models" / f"part-{k:05d}.avro")))
        assert len(a) == b.E and a == r
    res = json.load(open(moved / "result.json"))
    assert res["backend"] == backend
    rounds = res["rebalance"]
    assert rounds[0][0]["sent"][1] > 0 and rounds[0][0]["device"].startswith("cuda") and rounds[1][1]["device"].startswith("cuda")
