"""Worker of tests/test_strong_split.py: two gloo ranks run Rebalancer.exchange / give_back on CPU tensors with a measured-cost
model and an explicit travel order (rebalance.CostModel), with a stand-in "solve" whose result is a function of the entity's own
arrays — so that what comes back can be checked by position against the same function of the home copy."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gdmix_amd import synthetic  # noqa: E402
from gdmix_amd.rebalance import CostModel, Rebalancer, wire_tensors, wire_to_raw  # noqa: E402


def fake_solve(b):
    """Per entity: d+1 'coefficients' = [sum of values, then per distinct feature (ascending) the sum of its values], feature ids."""
    cc, th, fi = [], [], []
    for e in range(b.E):
        r0, r1 = b.ent_row_ptr[e], b.ent_row_ptr[e + 1]
        z0, z1 = b.row_nnz_ptr[r0], b.row_nnz_ptr[r1]
        cols, vals = b.col_global[z0:z1], b.val[z0:z1].astype(np.float64)
        u, inv = np.unique(cols, return_inverse=True)
        th.append(np.concatenate([[vals.sum() + b.y[r0:r1].sum()], np.bincount(inv, weights=vals, minlength=u.size)]))
        fi.append(u)
        cc.append(u.size + 1)
    return (np.array(cc, np.int64), np.concatenate(th) if th else np.zeros(0), np.concatenate(fi) if fi else np.zeros(0, np.int64))


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    # rank 0: many entities incl. three giants; rank 1: few
    b = synthetic.make_batch(400 if rank == 0 else 40, 20 if rank == 0 else 6, 4, 128, seed=11 + rank, size_dist="zipf" if rank == 0 else "poisson")
    n, z = b.ent_n(), b.ent_nnz()
    # three "classes" by size; the largest is not additive (must stay), the middle one costs most per non-zero (travels first)
    cls = np.where(z >= np.sort(z)[-3] if rank == 0 else np.zeros_like(z, bool), 2, np.where(z >= 64, 1, 0))
    class_ms = np.array([1.0, 6.0, 4.0]) * np.array([z[cls == c].sum() for c in range(3)]) / 1000.0
    tot = torch.from_numpy(CostModel.totals(cls, z, class_ms, 3))
    dist.all_reduce(tot)
    model = CostModel.from_totals(tot.numpy(), np.array([True, True, False]))
    cost, order = model.cost(cls, z), model.order(cls, z)
    rb = Rebalancer(n, z, wire_tensors(b, torch.device("cpu")), cost=cost, order=order)
    work = rb.exchange()
    wb = wire_to_raw(work)
    cc, th, fi = fake_solve(wb)
    ints = torch.arange(wb.E, dtype=torch.int32).reshape(-1, 1).repeat(1, 3)
    flts = torch.zeros((wb.E, 2), dtype=torch.float64)
    my_cc, my_th, _, my_fi, _, _ = rb.give_back(torch.from_numpy(cc), torch.from_numpy(th), None, torch.from_numpy(fi), ints, flts)
    hc, ht, hf = fake_solve(b)
    moved = np.concatenate(rb.sent)
    res = {"rank": rank, "E": b.E, "work_E": wb.E, "sent": [int(x.size) for x in rb.sent], "loads": rb.loads.tolist(),
           "cc_equal": bool(np.array_equal(my_cc.numpy(), hc)), "theta_equal": bool(np.array_equal(my_th.numpy(), ht)),
           "feat_equal": bool(np.array_equal(my_fi.numpy(), hf)),
           "moved_classes": sorted(set(cls[moved].tolist())), "moved_rate1_first": bool((cls[moved] == 1).sum() == min((cls == 1).sum(), moved.size)),
           "load_after": float(cost.sum() - cost[moved].sum()), "bytes_sent": rb.comm.bytes_sent, "bytes_received": rb.comm.bytes_received,
           "wire_released": rb.wire is None}
    allr = [None, None]
    dist.all_gather_object(allr, res)
    if rank == 0:
        json.dump(allr, open(out, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
