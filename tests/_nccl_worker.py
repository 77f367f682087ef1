"""Worker of the single-rank RCCL test (tests/test_rebalance.py, -m gpu): the RCCL branch of rebalance._Comm on HIP memory —
all_gather and variable-size all_to_all of every dtype the exchange uses, then a whole Rebalancer.exchange / give_back
round on the `nccl` backend, and the fixed-effect loop with its all-reduce on RCCL. One rank: the payloads travel rank 0 ->
rank 0 through RCCL on device tensors."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from gdmix_amd import synthetic
from gdmix_amd.rebalance import Rebalancer, _Comm, wire_tensors


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from gdmix_amd.solver import REDeviceSolver, SolverOptions
    solver = REDeviceSolver(0)
    dev = solver.device
    c = _Comm(device=dev)
    assert c.device.type == "cuda"
    assert np.array_equal(c.all_gather_floats(3.5), [3.5])
    rng = np.random.default_rng(0)
    for dtype, n in ((np.int64, 100_003), (np.float32, 70_001), (np.float64, 50_000), (np.uint8, 12_345), (np.int64, 0)):
        a = (rng.standard_normal(n) * 1000).astype(dtype)
        for _ in range(2):   # second round: staging buffers reused (the host side channel of the prior models)
            got = c.all_to_all([a], dtype)
            assert len(got) == 1 and got[0].dtype == dtype and np.array_equal(got[0], a), dtype
    x = torch.arange(12, dtype=torch.float64, device=dev).reshape(6, 2)
    assert torch.equal(c.all_to_all_t(x, [6], [6]), x) and np.array_equal(c.exchange_counts([[4, 5, 6]]), [[4, 5, 6]])
    # a whole round on device tensors: wire form up, exchange (one rank: nothing moves, every array still goes through RCCL),
    # widen + pack + solve of what came back, results through give_back, compared with the plain solve of the same batch
    b = synthetic.make_batch(500, 16, 4, 256, seed=4, size_dist="zipf")
    wire = wire_tensors(b, dev, solver)
    assert all(wire[k].is_cuda for k in ("ent_n", "row_nnz", "col_global", "val", "y", "offset"))
    rb = Rebalancer(b.ent_n(), b.ent_nnz(), wire)
    work = rb.exchange()
    assert work["E"] == b.E and work["Z"] == b.Z and all(work[k].is_cuda for k in ("ent_n", "row_nnz", "col_global", "val", "y", "offset"))
    assert np.array_equal(work["col_global"].cpu().numpy(), b.col_global)      # nothing to move on one rank
    opts = SolverOptions(regularize_bias=False)
    packed = solver.pack(solver.widen(work))
    solved = solver.solve(packed, opts)
    fp = packed.ent_feat_ptr()
    ints = torch.stack([solved.nit, solved.nfev, solved.status], dim=1)
    flts = torch.stack([solved.fval, solved.gnorm], dim=1)
    cc, th, va, fi, i2, f2 = rb.give_back((fp[1:] - fp[:-1]) + 1, solved.theta_thr, None, packed.unique_global(), ints, flts)
    assert th.is_cuda and fi.is_cuda and i2.is_cuda and va is None
    plain = solver.solve(solver.pack(b), opts).to_host()
    assert np.array_equal(th.cpu().numpy(), plain["theta_thr"]) and np.array_equal(i2[:, 0].cpu().numpy(), plain["nit"])
    assert np.array_equal(cc.cpu().numpy(), np.diff(packed.coef_ptr_host())) and np.array_equal(f2[:, 0].cpu().numpy(), plain["fval"])
    # the fixed-effect loop with its all-reduce on RCCL: [gradient, value] summed in place in the problem's device buffer between
    # gdmix_fe_eval and gdmix_fe_step (one rank: the sum is the identity, the coefficients are those of the plain loop, bitwise)
    from gdmix_amd import fixed_effect as fe
    n, k, D = 20_000, 6, 3_000
    rp = np.arange(n + 1, dtype=np.int64) * k
    cols = rng.integers(0, D, n * k)
    vals = rng.standard_normal(n * k).astype(np.float32)
    y = (rng.random(n) < 0.4).astype(np.float32)
    batch, _ = fe.shard_as_batch(rp, cols, vals, y, None, None, True)
    opts = SolverOptions(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=25, threshold=0.0, sum_loss=True)
    thetas = []
    def ordered(t):      # RCCL on the device buffer is ordered on the stream: the loop may run ahead of the status it has read
        dist.all_reduce(t)
    ordered.device_ordered = True
    for reduce, ahead in ((None, None), (lambda t: dist.all_reduce(t), 0), (ordered, None), (ordered, 3)):
        prob = fe._SteppingProblem(solver, solver.pack(batch, has_intercept=True), D, opts, None)
        assert prob.reduce_tensor().is_cuda
        fe.run_stepping_loop(prob, reduce, lookahead=ahead)
        thetas.append(prob.result()[0])
        prob.close()
    assert all(np.array_equal(thetas[0], t) for t in thetas[1:]) and np.abs(thetas[0]).max() > 0
    dist.barrier()
    dist.destroy_process_group()
    print("nccl single-rank exchange ok")


if __name__ == "__main__":
    main()
