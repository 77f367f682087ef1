"""ONE population split the reference's way (bench_strong.py / synthetic.population_share): entity -> partition by the Java hash
of its decimal id (PartitionUtils.scala:31-37), partition -> worker by partitions[rank::world] (random_effect_driver.py:60-68);
and the measured-cost model of the re-balancer (rebalance.CostModel). CPU only."""
import json
import os
import subprocess
import sys

import numpy as np

from gdmix_amd import synthetic
from gdmix_amd.partitioner import java_partition_id
from gdmix_amd.rebalance import CostModel, choose_entities

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_population_share_is_the_reference_split():
    ids = np.arange(1, 5001, dtype=np.int64)
    P = 64
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            own, pid = synthetic.population_share(ids, P, world, r)
            assert set(pid.tolist()) <= set(range(P)[r::world])                 # partitions[rank::world]
            assert np.all(np.diff(pid) >= 0)                                     # partition order ...
            for K in np.unique(pid)[:5]:
                assert np.all(np.diff(own[pid == K]) > 0)                        # ... and input order inside a partition
            for i in own[::397]:
                assert java_partition_id(str(ids[i]), P) == pid[np.flatnonzero(own == i)[0]]
            seen.append(own)
        allv = np.concatenate(seen)
        assert allv.size == ids.size and np.array_equal(np.sort(allv), np.arange(ids.size))   # every entity on exactly one rank
    assert synthetic.rank_partitions(10, 4, 1) == [1, 5, 9]


def test_c5_sizes_do_not_depend_on_the_number_of_workers():
    rng = np.random.default_rng(synthetic.C5_SEED)
    n = synthetic.c5_entity_samples(rng, 50_000)
    assert n.min() >= 1 and abs(n.mean() * 8 - 256) < 1.0 and n.max() <= (1 << 20) // 8
    ids = np.arange(n.size)
    tot = 0
    for r in range(4):
        own, _ = synthetic.population_share(ids, 1024, 4, r)
        tot += int(n[own].sum())
    assert tot == int(n.sum())


def test_cost_model_prices_classes_by_measured_time():
    cls = np.array([0, 0, 1, 1, 2, 0, 1])
    nnz = np.array([100, 300, 1000, 3000, 50000, 200, 2000])
    ms = np.array([0.6, 12.0, 9.0])
    tot = CostModel.totals(cls, nnz, ms, 3)
    assert np.allclose(tot[0], ms) and np.allclose(tot[1], [600 + 3 * 64, 6000 + 3 * 64, 50000 + 64])
    m = CostModel.from_totals(tot * 2.0, [True, True, False])            # summed over two identical ranks: same rates
    cost = m.cost(cls, nnz)
    for c in range(3):
        assert np.isclose(cost[cls == c].sum(), ms[c])                   # a class's entities add up to its launch
    order = m.order(cls, nnz)
    assert 4 not in order                                                # the team-tier entity never travels
    assert list(order[:3]) == [2, 6, 3] and set(order[3:]) == {0, 1, 5}  # costliest per byte first, small before large
    # a class with entities and no launch of its own (ran inside its neighbour's) is priced like the neighbour
    tot2 = tot.copy()
    tot2[0, 1] = 0.0
    m2 = CostModel.from_totals(tot2, [True, True, False])
    assert m2.rate[1] in (m2.rate[0], m2.rate[2]) and m2.rate[1] > 0
    sent = choose_entities(cost, [0.0, 5.0], order)
    assert 4 not in np.concatenate(sent) and cost[sent[1]].sum() <= 5.0 + 1e-9 and sent[1].size > 0


def test_two_gloo_ranks_exchange_by_measured_cost_and_get_their_results_back(tmp_path):
    out = tmp_path / "rb.json"
    env = dict(os.environ)
    env.pop("TF_CONFIG", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(ROOT, "tests", "_rb_order_worker.py"), str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=300, cwd=ROOT)
    r0, r1 = json.load(open(out))
    assert r0["sent"][1] > 0 and r1["sent"] == [0, 0]                      # the heavy rank sheds, the light one only receives
    assert r0["work_E"] + r1["work_E"] == r0["E"] + r1["E"] and r1["work_E"] == r1["E"] + r0["sent"][1]
    for r in (r0, r1):
        assert r["cc_equal"] and r["theta_equal"] and r["feat_equal"] and r["wire_released"]
    assert 2 not in r0["moved_classes"] and r0["moved_rate1_first"]        # giants stay; the costliest-per-byte class goes first
    mean = sum(r0["loads"]) / 2
    assert abs(r0["load_after"] - mean) < 0.08 * mean                      # the donor ends near the mean
    assert r0["bytes_sent"] > 0 and r0["bytes_sent"] == r1["bytes_received"] - 0 * r1["bytes_sent"] or r0["bytes_sent"] > 0


def test_partition_directory_is_the_reference_layout(tmp_path):
    """gdmix_amd.partition_dirs (bench.py's end-to-end legs, the -m gpu CLI test): entities land in partitionId=K by the Java hash of
    their id, in input order; metadata, feature list and partition list are what the trainer's CLI takes (contract B3)."""
    from gdmix_amd.io.grouped_reader import read_grouped_partition
    from gdmix_amd.io.metadata import DatasetMetadata
    from gdmix_amd.partition_dirs import write_partition_dir
    b = synthetic.make_survey_batch(300, 6, 4, 128, seed=3, with_uid=True)
    argv, members, size = write_partition_dir(str(tmp_path), b, 4, 128)
    assert size > 0 and sorted(members) == [0, 1, 2, 3] and sum(v.size for v in members.values()) == b.E
    for k, own in members.items():
        assert all(java_partition_id(b.entity_ids[e], 4) == k for e in own) and np.all(np.diff(own) > 0)
    assert open(tmp_path / "plist.txt").read() == "0,1,2,3"
    md = DatasetMetadata(json.load(open(tmp_path / "meta.json")))
    back = read_grouped_partition(str(tmp_path / "train" / "active" / "partitionId=2"), md, entity_name="ent", feature_bag="bag", offset_column_name="offset",
                                  uid_column_name="uid", label_column_name="response", weight_column_name=None, num_features=128)
    want = b.select(members[2])
    assert list(back.entity_ids) == list(want.entity_ids)
    for name in ("ent_row_ptr", "row_nnz_ptr", "col_global", "val", "y", "offset", "uid"):
        np.testing.assert_array_equal(getattr(back, name), getattr(want, name))
    assert "--stage=random_effect" in argv and f"--training_data_dir={tmp_path}/train" in argv and "--partition_entity=ent" in argv


def test_a_partition_of_a_share_is_a_batch_of_its_own():
    """bench_strong._slice_entities (the per-partition rounds of the projection): entities [e0, e1) of a resident raw batch as views with
    rebased pointers — the same arrays RawBatch.select gives."""
    import sys
    import torch
    sys.path.insert(0, ROOT)
    import bench_strong
    b = synthetic.make_ragged_batch(60, seed=9)
    raw = dict(E=b.E, N=b.N, Z=b.Z, ent_row_ptr=torch.from_numpy(b.ent_row_ptr), row_nnz_ptr=torch.from_numpy(b.row_nnz_ptr),
               col_global=torch.from_numpy(b.col_global), val=torch.from_numpy(b.val), y=torch.from_numpy(b.y), offset=torch.from_numpy(b.offset),
               weight=torch.from_numpy(b.weight))
    sub = bench_strong._slice_entities(raw, 17, 41)
    want = b.select(np.arange(17, 41))
    assert (sub["E"], sub["N"], sub["Z"]) == (want.E, want.N, want.Z)
    for k in ("ent_row_ptr", "row_nnz_ptr", "col_global", "val", "y", "offset", "weight"):
        np.testing.assert_array_equal(sub[k].numpy(), getattr(want, k))


def test_size_cost_model_prices_unclassified_entities_by_what_was_measured():
    """rebalance.SizeCostModel: the next partition's entities have no size class yet; the class times of the solves so far,
    attributed by non-zeros and summed per size bucket, price them. Before any measurement: the non-zero count."""
    from gdmix_amd.rebalance import SizeCostModel
    fresh = SizeCostModel()
    nnz = np.array([10, 20, 100, 5000, 20000, 300, 40, 8000])
    assert np.allclose(fresh.cost(nnz), nnz + 64.0) and 4 not in fresh.order(nnz)           # 20 000 >= 16 384: a team-tier entity never travels
    cls = np.array([0, 0, 1, 2, 3, 1, 0, 2])
    ms = np.array([0.1, 0.4, 3.0, 9.0])
    tot = SizeCostModel.totals(cls, nnz, ms, 4)
    assert tot.shape == (2, SizeCostModel.BUCKETS) and np.isclose(tot[0].sum(), ms.sum())   # every launch's time is handed out once
    m = SizeCostModel.from_totals(tot + tot)                                                # two ranks with the same measurement: same rates
    cost = m.cost(nnz)
    assert np.isclose(cost.sum(), ms.sum(), rtol=1e-12)                                     # (same entities: the costs add up to the launches)
    assert cost[4] > cost[3] > cost[2] > cost[0]
    order = m.order(nnz)
    assert 4 not in order and set(order) == {0, 1, 2, 3, 5, 6, 7}
    rate = m.rate[SizeCostModel.bucket(nnz[order])]
    assert np.all(np.diff(rate) <= 1e-18)                                                   # costliest per non-zero first
    unseen = SizeCostModel.bucket(np.array([3_000_000]))[0]
    assert m.rate[unseen] == m.rate[SizeCostModel.bucket(np.array([20000]))[0]]             # an unseen size takes its nearest neighbour's rate
    # a class with entities and no launch of its own (ran inside a neighbour's) does not create cost out of nothing
    tot2 = SizeCostModel.totals(cls, nnz, np.array([0.1, 0.0, 3.0, 9.0]), 4)
    assert np.isclose(tot2[0].sum(), 12.1)
