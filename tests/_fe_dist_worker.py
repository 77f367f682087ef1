"""Worker of the two-process fixed-effect test: each rank holds every other sample of a fixture as its shard and runs the
product path fit_stepping(). On a box with at least two GPUs every rank takes its own device and the all-reduce is RCCL on the
problem's device buffer (fixed_effect_lr_lbfgs_model.py:384-389 in the reference: two TF collectives); on the 1-GPU box the
ranks share GPU 0 and the all-reduce goes through gloo."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist

from gdmix_amd import fixed_effect as fe


def wide_case():
    """3 000 samples x 6 non-zeros over 4 500 features (a few of them in no sample at all), weights, offsets."""
    rng = np.random.default_rng(12)
    n, k, D = 3000, 6, 4500
    col = rng.integers(0, D - 40, (n, k)).astype(np.int64).ravel()      # the last 40 features never occur
    val = rng.standard_normal(n * k).astype(np.float32)
    y = (rng.random(n) < 0.5).astype(np.float32)
    off = (0.2 * rng.standard_normal(n)).astype(np.float32)
    wt = (0.5 + rng.random(n)).astype(np.float32)
    return np.arange(n + 1, dtype=np.int64) * k, col, val, y, off, wt, D


def main():
    base, names = sys.argv[1], sys.argv[2].split(",")
    world = int(os.environ["WORLD_SIZE"])
    rccl = torch.cuda.device_count() >= world
    dev = int(os.environ.get("LOCAL_RANK", "0")) if rccl else 0
    torch.cuda.set_device(dev)
    if rccl:
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    out = {"_backend": dist.get_backend(), "_device": dev}
    s = fe.FixedEffectDeviceSolver(dev)
    for name in names:
        z = np.load(os.path.join(ROOT, "tests", "golden", f"fe_{name}.npz"))
        rp, col, val = z["row_nnz_ptr"], z["col_global"], z["val"]
        rows = np.arange(rank, rp.size - 1, world)
        k = np.diff(rp)[rows]
        nz = np.concatenate([np.arange(rp[i], rp[i + 1]) for i in rows]) if rows.size else np.zeros(0, np.int64)
        theta, info = s.fit_stepping(np.concatenate([[0], np.cumsum(k)]), col[nz], val[nz], z["y"][rows], int(z["num_features"]),
                                     offset=z["offset"][rows], has_intercept=bool(z["has_intercept"]), l2=float(z["l2"]),
                                     regularize_bias=True,
                                     model_type=fe.LINEAR_REGRESSION if z["linear"] else fe.LOGISTIC_REGRESSION,
                                     theta0=z["theta0"] if z["theta0"].size else None, max_iter=int(z["max_iter"]))
        out[name] = {"theta": theta.tolist(), "status": int(info["status"]), "nit": int(info["nit"]), "nfev": int(info["nfev"])}
    if len(sys.argv) > 3 and sys.argv[3] == "full_variance":
        # FULL variances of a model too wide for the host path (P = 4 501 > 4 096): dense curvature matrix per worker on its device,
        # one all-reduce of the P x P matrix, replicated factorisation (fixed_effect._full_variances_several_workers)
        rp, col, val, y, off, wt, D = wide_case()
        rows = np.arange(rank, rp.size - 1, world)
        k = np.diff(rp)[rows]
        nz = np.concatenate([np.arange(rp[i], rp[i + 1]) for i in rows])
        theta, info = s.fit_stepping(np.concatenate([[0], np.cumsum(k)]), col[nz], val[nz], y[rows], D, offset=off[rows], weight=wt[rows],
                                     has_intercept=True, l2=0.7, regularize_bias=False, max_iter=4, variance_mode="FULL")
        out["full_variance"] = {"theta": theta.tolist(), "variances": np.asarray(info["variances"]).tolist()}
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        json.dump(gathered, open(os.path.join(base, "result.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
