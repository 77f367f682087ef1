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
from gdmix_amd.rebalance import Rebalancer, _Comm


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    c = _Comm()
    assert c.device.type == "cuda"
    assert np.array_equal(c.all_gather_floats(3.5), [3.5])
    rng = np.random.default_rng(0)
    for dtype, n in ((np.int64, 100_003), (np.float32, 70_001), (np.float64, 50_000), (np.uint8, 12_345), (np.int64, 0)):
        a = (rng.standard_normal(n) * 1000).astype(dtype)
        for _ in range(2):   # second round: staging buffers reused
            got = c.all_to_all([a], dtype)
            assert len(got) == 1 and got[0].dtype == dtype and np.array_equal(got[0], a), dtype
    b = synthetic.make_batch(500, 16, 4, 256, seed=4, size_dist="zipf")
    rb = Rebalancer(b)
    work = rb.exchange()
    assert work.E == b.E and np.array_equal(work.col_global, b.col_global)      # nothing to move on one rank
    coef_cnt = np.diff(b.ent_row_ptr) % 7 + 1
    theta = rng.standard_normal(int(coef_cnt.sum()))
    feat_cnt = coef_cnt - 1
    feat_idx = rng.integers(0, 256, int(feat_cnt.sum()))
    cc, th, va, fc, fi, st = rb.give_back(coef_cnt, theta, None, feat_cnt, feat_idx, {"nit": np.arange(b.E)})
    assert np.array_equal(cc, coef_cnt) and np.array_equal(th, theta) and va is None and np.array_equal(fi, feat_idx)
    assert np.array_equal(st["nit"], np.arange(b.E))
    # the fixed-effect loop with its all-reduce on RCCL: [gradient, value] summed in place in the problem's device buffer between
    # gdmix_fe_eval and gdmix_fe_step (one rank: the sum is the identity, the coefficients are those of the plain loop, bitwise)
    from gdmix_amd import fixed_effect as fe
    from gdmix_amd.solver import REDeviceSolver, SolverOptions
    n, k, D = 20_000, 6, 3_000
    rp = np.arange(n + 1, dtype=np.int64) * k
    cols = rng.integers(0, D, n * k)
    vals = rng.standard_normal(n * k).astype(np.float32)
    y = (rng.random(n) < 0.4).astype(np.float32)
    solver = REDeviceSolver(0)
    batch, _ = fe.shard_as_batch(rp, cols, vals, y, None, None, True)
    opts = SolverOptions(l2=1.0, regularize_bias=False, has_intercept=True, m=10, max_iter=25, threshold=0.0, sum_loss=True)
    thetas = []
    for reduce in (None, lambda t: dist.all_reduce(t)):
        prob = fe._SteppingProblem(solver, solver.pack(batch, has_intercept=True), D, opts, None)
        assert prob.reduce_tensor().is_cuda
        fe.run_stepping_loop(prob, reduce)
        thetas.append(prob.result()[0])
        prob.close()
    assert np.array_equal(thetas[0], thetas[1]) and np.abs(thetas[0]).max() > 0
    dist.barrier()
    dist.destroy_process_group()
    print("nccl single-rank exchange ok")


if __name__ == "__main__":
    main()
