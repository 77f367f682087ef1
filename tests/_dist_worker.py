"""Worker of the world_size-2 gloo test: each rank runs the RandomEffectDriver over a shared partition list
with the oracle-backed solver double and reports which partitions it trained."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch
import torch.distributed as dist

from helpers import OracleSolverDouble
from gdmix_amd.driver import RandomEffectDriver
from gdmix_amd.model import RandomEffectLRLBFGSModel
from gdmix_amd.params import Params, SchemaParams


def main():
    base = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])   # the driver opens the process group when it needs one
    argv = json.load(open(os.path.join(base, "argv.json")))
    model = RandomEffectLRLBFGSModel(argv)
    real = os.environ.get("GDMIX_TEST_DEVICE_SOLVER") == "1"   # -m gpu on a box with a GPU per rank: the product path, RCCL
    if real:
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count())   # (1-GPU box: the ranks share it, GDMIX_RANKS_SHARE_DEVICE)
    else:
        model._solver = OracleSolverDouble()
    # record what every re-balancing round did on this rank
    rounds = []
    from gdmix_amd import rebalance as rbm
    orig_exchange = rbm.Rebalancer.exchange

    def exchange(self, *a, **k):
        work = orig_exchange(self, *a, **k)
        rounds.append({"entities": int(self.E), "sent": [int(x.size) for x in self.sent], "received": list(self.recv_counts),
                       "solved": int(work["E"]), "with_prior": bool(k.get("with_prior", False)),
                       "prior_models": int(self.work_prior["has"].sum()) if self.work_prior is not None else 0,
                       "device": str(work["val"].device)})
        return work
    rbm.Rebalancer.exchange = exchange
    driver = RandomEffectDriver(Params.__from_argv__(argv), model)
    assert driver.execution_context["task_index"] == rank and driver.execution_context["num_workers"] == world
    mine = driver._get_partition_list()
    driver.run_training(SchemaParams.__from_argv__(argv), export_model=True)
    if not dist.is_initialized():
        dist.init_process_group("gloo")
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    # load totals: the only collective the RE path needs is this kind of tiny metadata exchange
    t = torch.tensor([len(mine)], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t)
    all_rounds = [None] * world
    dist.all_gather_object(all_rounds, rounds)
    if rank == 0:
        json.dump({"per_rank": gathered, "total": int(t.item()), "rebalance": all_rounds, "backend": dist.get_backend()},
                  open(os.path.join(base, "result.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
