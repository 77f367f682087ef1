"""ModelTable (gdmix_amd/model.py) answers through native joins of the id bytes as long as nobody needs the dict
(random_effect_lr_lbfgs_model.py:141-162: dict.update of the trained models over the prior ones). Every such answer must be the
dict's: the same table built from plain lists of str takes the dict path."""
import numpy as np
import pytest

from gdmix_amd.io import native_reader
from gdmix_amd.io.native_reader import EntityIds, _ids_to_bytes
from gdmix_amd.model import ModelTable

pytestmark = pytest.mark.skipif(not native_reader.available(), reason="libgdmix_io.so is not built")


def _native_ids(ids):
    joined = [s.encode("utf-8") for s in ids]
    ptr = np.zeros(len(ids) + 1, np.int64)
    np.cumsum([len(x) for x in joined], out=ptr[1:])
    return EntityIds(b"".join(joined), ptr)


def _chunk(rng, ids, with_var):
    p = rng.integers(1, 6, len(ids))
    coef_ptr = np.concatenate([[0], np.cumsum(p)]).astype(np.int64)
    feat_ptr = np.concatenate([[0], np.cumsum(p - 1)]).astype(np.int64)
    theta = rng.standard_normal(int(coef_ptr[-1]))
    idx = rng.integers(0, 100, int(feat_ptr[-1])).astype(np.int64)
    var = rng.random(int(coef_ptr[-1])) if with_var else None
    return theta, coef_ptr, idx, feat_ptr, var


def _tables(rng, id_lists, with_var):
    fast, slow = ModelTable(), ModelTable()
    for k, ids in enumerate(id_lists):
        ch = _chunk(rng, ids, with_var[k])
        f, s = ModelTable(), ModelTable()
        f.add_chunk(_native_ids(ids), *ch)
        s.add_chunk(list(ids), *ch)
        fast.update(f)
        slow.update(s)
    return fast, slow


def _same_flat(a, b):
    assert list(a[0]) == list(b[0])
    for x, y in zip(a[1:], b[1:]):
        assert (x is None) == (y is None)
        if x is not None:
            np.testing.assert_array_equal(np.asarray(x), np.asarray(y))


@pytest.mark.parametrize("shape", ["one chunk", "prior + trained", "three chunks", "disjoint", "identical ids"])
@pytest.mark.parametrize("with_var", [(False, False, False), (True, False, True)])
def test_native_answers_equal_the_dicts(shape, with_var):
    rng = np.random.default_rng(5)
    base = [f"e{i}" for i in rng.permutation(400)] + ["", "ü-umlaut", "0", "00"]
    trained = [base[i] for i in rng.permutation(len(base))[:250]] + [f"new{i}" for i in range(60)]
    rng.shuffle(trained)
    lists = {"one chunk": [base], "prior + trained": [base, trained], "three chunks": [base, trained, base[:50] + ["zz"]],
             "disjoint": [base, [f"n{i}" for i in range(30)]], "identical ids": [base, list(base)]}[shape]
    fast, slow = _tables(np.random.default_rng(7), lists, with_var)
    assert fast._where == {} and (len(lists) > 2 or fast._plan() is not None)
    assert len(fast) == len(slow) and bool(fast) == bool(slow)
    _same_flat(fast.flatten(), slow.flatten())
    # lookups: ids of the table, ids it does not have, in another order
    q = [lists[-1][i] for i in rng.permutation(len(lists[-1]))[:40]] + ["nobody", ""] + lists[0][:25]
    for a, b in zip(fast.lookup(_native_ids(q)), slow.lookup(q)):
        np.testing.assert_array_equal(a, b)
    assert fast._where == {} or len(lists) > 2       # (three chunks: flatten took the dict path)
    rf, rs = fast.rows_for(_native_ids(q)), slow.rows_for(q)
    for k in rf:
        np.testing.assert_array_equal(rf[k], rs[k])
    assert ("e3" in fast) == ("e3" in slow) and fast._where     # now it is a dict, and still the same table
    _same_flat(fast.flatten(), slow.flatten())


def test_repeated_ids_take_the_dict_path():
    rng = np.random.default_rng(1)
    ids = ["a", "b", "a", "c"]
    ch = _chunk(rng, ids, False)
    fast, slow = ModelTable(), ModelTable()
    fast.add_chunk(_native_ids(ids), *ch)
    slow.add_chunk(ids, *ch)
    assert not _native_ids(ids).all_different() and fast._plan() is None
    assert len(fast) == len(slow) == 3
    _same_flat(fast.flatten(), slow.flatten())


def test_entity_ids_is_the_list_it_stands_for():
    ids = ["10", "", "äö", "7" * 40]
    e = _native_ids(ids)
    assert len(e) == 4 and list(e) == ids and e == ids and e[2] == "äö" and e[1:3] == ids[1:3]
    raw, ptr = _ids_to_bytes(e)
    raw2, ptr2 = _ids_to_bytes(ids)
    assert raw == raw2 and np.array_equal(ptr, ptr2)
    assert list(e.take([3, 0])) == [ids[3], ids[0]] and list(e.extended(e.take([1]))) == ids + [""]
    np.testing.assert_array_equal(_native_ids(["7" * 40, "x", "10"]).rows_in(e), [3, -1, 0])
