"""gdmix_amd/partitioner.py against the known answers of the reference's own Scala tests
(gdmix-data/src/test/scala/com/linkedin/gdmix/data/{OffsetUpdaterTest,DataPartitionerTest}.scala: data lifted as
vectors) and the hand-derived Java-hash KATs of SURVEY.md §8(b) B4. CPU only."""
import os

import numpy as np
import pytest

from gdmix_amd import partitioner as pt
from gdmix_amd.io.grouped_reader import read_grouped_partition
from oracle import oracle

# DataPartitionerTest.scala:25-32
UID = np.arange(10, dtype=np.int64)
ENTITY = np.array([0, 0, 0, 1, 1, 1, 1, 1, 1, 2], np.int64)
LABEL = np.array([0, 0, 1, 1, 1, 0, 0, 1, 1, 1], np.float32)
INDICES = [[0, 1], [0, 1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [3, 4], [5, 9], [0], [0, 2]]
VALUES = [[0, 1], [0, 1.0, 2.2], [3, 4.1], [5.5, 6.6], [7.7, 8.8], [9.3, 10.12], [0.3, 0.8], [0.8, 1.8], [0.0], [1.0, -2.2]]
# (the reference's ninth value list has two entries for one index; one is kept so that the bag stays well-formed)
RNP = np.concatenate([[0], np.cumsum([len(x) for x in INDICES])]).astype(np.int64)
COLS = np.concatenate(INDICES).astype(np.int64)
VALS = np.concatenate(VALUES).astype(np.float32)

MD = {"features": [{"name": "global", "dtype": "float", "shape": [16], "isSparse": True},
                   {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                   {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                   {"name": "entityId", "dtype": "long", "shape": [], "isSparse": False}],
      "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}


def test_offset_updater_known_answers():
    # OffsetUpdaterTest.scala:27-58
    rows, off = pt.update_offsets([1, 2], [1, 2], [1.0, 2.0])
    assert rows.tolist() == [0, 1] and off.tolist() == [1.0, 2.0]
    rows, off = pt.update_offsets([1, 2], [1, 2], [1.0, 2.0], [1, 2], [0.1, 0.2])
    assert off.dtype == np.float32
    assert off.tolist() == [np.float32(0.9), np.float32(1.8)]
    # inner joins: rows without a previous score (or without a per-coordinate score) are dropped; order of `uid` kept
    rows, off = pt.update_offsets([5, 2, 9, 1], [1, 2, 9], [1.0, 2.0, 9.0], [9, 2], [0.5, 0.25])
    assert rows.tolist() == [1, 2] and off.tolist() == [1.75, 8.5]


def test_group_ids_follow_the_reference_test():
    # DataPartitionerTest.scala:100-120: lowerBound 2, upperBound 4
    gid = pt.group_ids(ENTITY, UID, 2, 4)
    assert set(gid[ENTITY == 0].tolist()) == {0}              # 3 samples: one group, active
    assert set(gid[ENTITY == 1].tolist()) == {0, 1}           # 6 samples: 6 / 4 + 1 = 2 groups by pmod(uid, 2)
    assert gid[ENTITY == 1].tolist() == [int(u % 2) for u in UID[ENTITY == 1]]
    assert set(gid[ENTITY == 2].tolist()) == {-1}             # 1 sample < lower bound: passive
    assert not pt.group_ids(ENTITY, UID).any()
    assert pt.group_ids(ENTITY, -UID - 1, None, 4)[3:9].tolist() == [int((-u - 1) % 2) for u in UID[3:9]]   # pmod is non-negative


def test_bound_and_group_matches_the_reference_expectations():
    # DataPartitionerTest.scala:35-46,150-190 (no bounds: one record per entity, samples in input order)
    order, ptr, ent, gid = pt.group_samples(ENTITY, pt.group_ids(ENTITY, UID))
    assert ent.tolist() == [0, 1, 2] and gid.tolist() == [0, 0, 0]
    groups = [UID[order[ptr[r]:ptr[r + 1]]].tolist() for r in range(3)]
    assert groups == [[0, 1, 2], [3, 4, 5, 6, 7, 8], [9]]
    assert [LABEL[order[ptr[r]:ptr[r + 1]]].astype(int).tolist() for r in range(3)] == [[0, 0, 1], [1, 1, 0, 0, 1, 1], [1]]


def test_java_hash_known_answers():
    kats = {"0": 48, "100034": 1448635136, "abc102": -1424436655, "polygenelubricants": -2147483648, "Aa": 2112, "BB": 2112,
            "\U0001F600": 1772899}
    for s, h in kats.items():
        assert pt.java_string_hash(s) == h, s
    assert pt.java_partition_id("polygenelubricants", 10) == -8          # abs(Int.MinValue) stays negative, % keeps the sign
    assert [pt.java_partition_id(s, 10) for s in ("0", "1", "12", "100", "943", "1682")] == [8, 9, 9, 5, 0, 9]
    rng = np.random.default_rng(0)
    ids = np.concatenate([rng.integers(-2 ** 63, 2 ** 63 - 1, 3000), rng.integers(-1000, 1000, 1000),
                          [0, -1, 2 ** 63 - 1, -2 ** 63, 10 ** 18, -10 ** 18]]).astype(np.int64)
    for n in (1, 7, 10, 1024):
        vec = pt.java_partition_ids_int64(ids, n)
        assert vec.tolist() == [pt.java_partition_id(int(x), n) for x in ids]
        assert vec[:300].tolist() == [oracle.java_partition_id(str(int(x)), n) for x in ids[:300]]


@pytest.mark.parametrize("bounds", [(None, None), (2, 4)])
def test_partitions_on_disk_round_trip(tmp_path, bounds):
    lb, ub = bounds
    batches = pt.build_batches(ENTITY, UID, LABEL, np.linspace(0, 1, 10).astype(np.float32), None, RNP, COLS, VALS, 3, lb, ub)
    paths = pt.write_partitions(str(tmp_path), batches, "entityId", "global", int_entity_ids=True, weight_column_name=None)
    assert all(os.path.exists(p) for p in paths)
    seen_uid = []
    for (sub, p), b in batches.items():
        assert sub in ("active", "passive")
        d = os.path.join(str(tmp_path), sub, f"partitionId={p}")
        r = read_grouped_partition(d, MD, "entityId", "global", "offset", "uid", "response", None, num_features=16)
        assert r.entity_ids == b.entity_ids
        np.testing.assert_array_equal(r.uid, b.uid)
        np.testing.assert_array_equal(r.col_global, b.col_global)
        np.testing.assert_array_equal(r.val, b.val)
        for e in r.entity_ids:
            assert pt.java_partition_id(e, 3) == p
        seen_uid += r.uid.tolist()
        if sub == "active" and lb is not None:
            assert "2" not in r.entity_ids                    # the single-sample entity is passive
    assert sorted(seen_uid) == UID.tolist()                   # every sample lands exactly once
    if lb is None:
        assert all(sub == "active" for sub, _ in batches)     # no bounds: no passive data is written
    # feature rows travel with their samples
    b_all = pt.build_batches(ENTITY, UID, LABEL, np.zeros(10, np.float32), None, RNP, COLS, VALS, 1, split=False)
    b0 = b_all[("", 0)]
    assert b0.entity_ids == ["0", "1", "2"] and b0.col_global.tolist() == COLS.tolist()
