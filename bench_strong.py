"""Strong-scaling legs of bench.py: ONE population split over the ranks the reference's way, and the re-balancer timed.

BASELINE.json configs[2] is one MovieLens-20M split over 8 GPUs and configs[4] one 100 M-entity Zipf population "with dynamic entity
rebalancing" — not eight copies. The split is the reference's: entity -> partition by the Java hash of its decimal id
(gdmix-data/.../PartitionUtils.scala:31-37, contract B4: gdmix_java_partition_ids_i64 on the device), partition -> worker by
partitions[rank::world] (gdmix-trainer/src/gdmix/drivers/random_effect_driver.py:60-68). A rank's share is its partitions
concatenated in partition order; a step is pack + solve of the share, resident in HBM, exactly as in the weak-scaling headline.

  strong_leg        N ranks (torch.distributed): every rank generates its share of the population, K timed steps between barriers,
                    job time = the slowest rank. With `rebalance` the share then goes through gdmix_amd.rebalance: cost model from
                    the plain run's per-class kernel times (CostModel), exchange -> widen -> pack -> solve -> give_back, every phase
                    timed, bytes sent / received counted, and the coefficients that come back compared with the plain run's.
  projection_leg    one process, one GPU: the R shares solved one after another on the same device. Entities are independent and
                    the data path has no collective, so max over shares is what an R-GPU job would take apart from its barriers;
                    this is how a 1-GPU box can say something about ML-20M over 8 GPUs (17 k users per GPU: a latency-bound regime).
                    Labelled as a projection wherever it is reported; it also reports the re-balancing *plan* (predicted loads
                    before / after, bytes that would move) without executing it.
"""
import time

import numpy as np

STRONG_WORKLOADS = {
    "ml20m_user": ("C3: ONE MovieLens-20M per-user population (138 493 users, 16 M training rows) hashed into 64 partitions", 64),
    "ml20m_movie": ("C3: ONE MovieLens-20M per-movie population (26 744 movies, 16 M training rows) hashed into 64 partitions", 64),
    "c5": ("C5: ONE Zipf population of --c5-entities x ranks entities (P(nnz >= x) ~ x^-1.2 on [8, 2^20], mean 256, D = 65 536) "
           "hashed into 1 024 partitions (SURVEY 8(d))", 1024),
}


class Share:
    """A rank's share of a population, raw form resident in HBM."""

    def __init__(self, name, raw_dev, n, z, ids, pid, total_entities, partitions, gen_s):
        self.name, self.raw_dev, self.n, self.z, self.ids, self.pid = name, raw_dev, n, z, ids, pid
        self.E, self.N, self.Z = int(n.size), int(n.sum()), int(z.sum())
        self.total_entities, self.partitions, self.gen_s = int(total_entities), int(partitions), gen_s


_ML_CACHE = {}


def ml20m_population(kind, ml_entities=None):
    """The MovieLens-20M-sized population (seed 200), generated once per process: the projection asks for every share of it."""
    from gdmix_amd import synthetic
    key = (kind, ml_entities or None)
    if key not in _ML_CACHE:
        _ML_CACHE[key] = synthetic.make_movielens_20m(kind, seed=200, entities=ml_entities or None)
    return _ML_CACHE[key]


def make_share(name, world, rank, solver, c5_entities, ml_entities=None):
    """Rank `rank`'s share of population `name` for a job of `world` workers."""
    from gdmix_amd import synthetic
    t0 = time.perf_counter()
    P = STRONG_WORKLOADS[name][1]
    dev_pids = lambda ids, parts: solver.partition_ids(np.ascontiguousarray(ids, np.int64), parts).cpu().numpy()
    if name == "c5":
        total = int(c5_entities) * world
        raw, n, ids, pid = synthetic.make_c5_population_share(solver.device, total, world, rank, P, partition_ids_fn=dev_pids)
        z = n * (raw["Z"] // max(1, raw["N"]))
        return Share(name, raw, n, z, ids, pid, total, P, time.perf_counter() - t0)
    full = ml20m_population("per_user" if name == "ml20m_user" else "per_movie", ml_entities)
    ids_all = np.arange(1, full.E + 1, dtype=np.int64)          # MovieLens ids are 1-based decimals
    own, pid = synthetic.population_share(ids_all, P, world, rank, dev_pids(ids_all, P))
    sub = full.select(own)
    return Share(name, solver.upload(sub), sub.ent_n(), sub.ent_nnz(), ids_all[own], pid, full.E, P, time.perf_counter() - t0)


def _wire_from_raw(raw):
    """The 32-bit wire form (what the re-balancer exchanges) of a raw batch in HBM."""
    import torch
    d = dict(E=raw["E"], N=raw["N"], Z=raw["Z"], row_nnz_width=4, y_width=4, col_width=4)
    d["ent_n"] = (raw["ent_row_ptr"][1:] - raw["ent_row_ptr"][:-1]).to(torch.int32)
    d["row_nnz"] = (raw["row_nnz_ptr"][1:] - raw["row_nnz_ptr"][:-1]).to(torch.int32)
    d["col_global"] = raw["col_global"].to(torch.int32)
    d["val"], d["y"], d["offset"], d["weight"] = raw["val"], raw["y"], raw["offset"], raw["weight"]
    return d


def _timed_steps(solver, share, opts, steps, warmup, sync, barrier):
    """-> (packed, res, seconds of this rank's `steps` steps, seconds until every rank is done, class ms of the last step)."""
    packed = res = None
    for _ in range(max(1, warmup)):
        packed = res = None
        packed = solver.pack(share.raw_dev)
        res = solver.solve(packed, opts)
    sync()
    barrier()
    import gc
    import torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    pack_ms = 0.0
    # a share's step is 2 ms: one full collection of the interpreter's garbage (it came at the same call count of a run, whatever the
    # share: + 6 - 10 ms on ONE share of eight, tools/r04_tallteam2.sh) would be the slowest "rank" of the job — collected before, not during
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    share.step_ms = []
    try:
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            packed = res = None
            ev[0].record()
            packed = solver.pack(share.raw_dev)
            ev[1].record()
            res = solver.solve(packed, opts)
            ev[2].record()
            ev[2].synchronize()
            pack_ms += ev[0].elapsed_time(ev[1])
            share.step_ms.append(round((time.perf_counter() - t1) * 1e3, 3))
        sync()
        own = time.perf_counter() - t0
    finally:
        if gc_was_on:
            gc.enable()
    barrier()
    share.pack_ms = pack_ms / steps
    return packed, res, own, time.perf_counter() - t0, np.array(solver.last_solve_ms())


def _converged(res):
    st = res.status
    return int(((st >= 0) & (st <= 2)).sum().item())


def _entity_classes(packed, torch):
    return packed._view(packed.c.cls_tmp, packed.E, torch.int32).cpu().numpy()


def strong_leg(name, rank, world, solver, opts, coll_dev, c5_entities, steps=2, warmup=1, rebalance=True, ml_entities=None, tolerance=0.05, rounds=4):
    """Every rank: its share of population `name`, `steps` timed steps. -> dict on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    from gdmix_amd.rebalance import CostModel, Rebalancer, _Comm
    from gdmix_amd.solver import NUM_CLASSES
    sync = torch.cuda.synchronize
    barrier = (lambda: dist.barrier()) if world > 1 else (lambda: None)

    def gather_rows(row):
        mine = torch.tensor([float(v) for v in row], dtype=torch.float64, device=coll_dev)
        if world == 1:
            return mine.cpu().numpy()[None, :]
        out = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(out, mine)
        return np.stack([o.cpu().numpy() for o in out])

    share = make_share(name, world, rank, solver, c5_entities, ml_entities)
    solver.set_timing(True)
    packed, res, own_s, job_s, class_ms = _timed_steps(solver, share, opts, steps, warmup, sync, barrier)
    conv = _converged(res)
    rows = gather_rows([rank, share.E, share.N, share.Z, own_s / steps * 1e3, conv, job_s, float(class_ms.sum()), share.gen_s,
                        len(set(share.pid.tolist())), share.pack_ms])
    job_ms = float(rows[:, 6].max()) / steps * 1e3
    ms = rows[:, 4]
    out = {"workload": name, "what": STRONG_WORKLOADS[name][0], "partitions": share.partitions, "ranks": world,
           "split": "entity -> partition: Java hash of the decimal id (PartitionUtils.scala:31-37); partition -> rank: partitions[rank::ranks] "
                    "(random_effect_driver.py:60-68)",
           "total_entities": int(rows[:, 1].sum()), "total_nnz": int(rows[:, 3].sum()), "steps": steps,
           "ms": job_ms, "entities_per_s": float(rows[:, 5].sum()) / (job_ms * 1e-3), "converged_per_step": int(rows[:, 5].sum()),
           "imbalance": float(ms.max() / ms.mean()),
           "per_rank": [{"rank": int(r[0]), "entities": int(r[1]), "samples": int(r[2]), "nnz": int(r[3]), "ms_per_step": float(r[4]),
                         "pack_ms": float(r[10]), "solve_kernel_ms": float(r[7]), "partitions": int(r[9]), "generate_s": round(float(r[8]), 2)}
                        for r in rows]}
    assert out["total_entities"] == share.total_entities, (out["total_entities"], share.total_entities)
    if rebalance and world > 1:
        # ---- the same share through the re-balancer ----
        cls = _entity_classes(packed, torch)
        additive = np.arange(NUM_CLASSES) <= NUM_CLASSES - 5          # everything but the multi-workgroup team tiers
        tot = torch.from_numpy(CostModel.totals(cls, share.z, class_ms, NUM_CLASSES)).to(coll_dev)
        dist.all_reduce(tot)
        model = CostModel.from_totals(tot.cpu().numpy(), additive)
        cost, order = model.cost(cls, share.z), model.order(cls, share.z)
        theta_plain, status_plain = res.theta_thr.clone(), res.status.clone()
        cp_plain = torch.from_numpy(packed.coef_ptr_host()).to(solver.device)
        packed = res = None
        torch.cuda.empty_cache()
        wire = _wire_from_raw(share.raw_dev)
        comm = _Comm(None, solver.device)
        phases = np.zeros(5)
        moved = None
        diff = 0.0
        status_equal = 0
        for it in range(warmup + steps):
            if it == warmup:
                phases[:] = 0
                comm.bytes_sent = comm.bytes_received = 0
                sync(); barrier()
                t_job = time.perf_counter()
            t = [time.perf_counter()]
            rb = Rebalancer(share.n, share.z, dict(wire), cost=cost, order=order, comm=comm, tolerance=tolerance)
            work = rb.exchange()
            sync(); t.append(time.perf_counter())
            pk = solver.pack(solver.widen(work))
            sync(); t.append(time.perf_counter())
            rs = solver.solve(pk, opts)
            sync(); t.append(time.perf_counter())
            fp = pk.ent_feat_ptr()
            cc = (fp[1:] - fp[:-1]) + 1
            ints = torch.stack([rs.nit, rs.nfev, rs.status], dim=1)
            flts = torch.stack([rs.fval, rs.gnorm], dim=1)
            my_cc, th, _, _, my_ints, _ = rb.give_back(cc, rs.theta_thr, None, pk.unique_global(), ints, flts)
            sync(); t.append(time.perf_counter())
            barrier(); t.append(time.perf_counter())
            phases += np.diff(t)
            moved = [int(ix.size) for ix in rb.sent]
            if it == warmup + steps - 1:
                # what came back, against the plain run of the same entities (the multi-workgroup team tiers reproduce to ~1e-8:
                # a tier's team size depends on what else is in it; everything else is bit for bit)
                assert th.numel() == theta_plain.numel() and bool((my_cc.cumsum(0) == cp_plain[1:]).all())
                scale = torch.clamp(theta_plain.abs(), min=1e-3)
                diff = float(((th - theta_plain).abs() / scale).max().item()) if th.numel() else 0.0
                status_equal = int((my_ints[:, 2] == status_plain).sum().item())
            del rb, work, pk, rs, th
        job2 = time.perf_counter() - t_job
        rrows = gather_rows([rank, *(phases / steps * 1e3), comm.bytes_sent / steps, comm.bytes_received / steps, sum(moved), diff, status_equal,
                             float(cost.sum()), job2])
        job2_ms = float(rrows[:, 12].max()) / steps * 1e3
        work_ms = rrows[:, 2] + rrows[:, 3]        # pack + solve where the entities ended up
        out["rebalanced"] = {
            "ms": job2_ms, "entities_per_s": out["converged_per_step"] / (job2_ms * 1e-3),
            "imbalance_of_pack_plus_solve": float(work_ms.max() / work_ms.mean()), "tolerance": tolerance,
            "cost_model": "milliseconds: per-class kernel time / non-zeros of the plain run, summed over ranks (rebalance.CostModel); entities of the "
                          "multi-workgroup team tiers stay; the costliest per byte travel first",
            "max_rel_diff_vs_plain": float(rrows[:, 9].max()), "status_equal": int(rrows[:, 10].sum()),
            "per_rank": [{"rank": int(r[0]), "exchange_ms": float(r[1]), "widen_pack_ms": float(r[2]), "solve_ms": float(r[3]), "give_back_ms": float(r[4]),
                          "wait_ms": float(r[5]), "bytes_sent": int(r[6]), "bytes_received": int(r[7]), "entities_sent": int(r[8]),
                          "predicted_cost_ms": float(r[11])} for r in rrows]}
    if name == "c5" and rounds > 0 and world > 1:
        out["partition_rounds"] = _rounds_leg(share, rank, world, solver, opts, coll_dev, rounds, rebalance, tolerance, gather_rows)
    return out if rank == 0 else None


def _rounds_leg(share, rank, world, solver, opts, coll_dev, rounds, rebalance, tolerance, gather_rows):
    """The product path's granularity with ranks: every worker trains ONE partition per round, in lockstep (model.py with
    --rebalance_entities). Per round: the plain pack + solve of each worker's partition, then the same round through the re-balancer,
    whose cost model (rebalance.SizeCostModel) is what the rounds before it measured — the first round prices by non-zeros."""
    import torch
    import torch.distributed as dist
    from gdmix_amd.rebalance import Rebalancer, SizeCostModel, _Comm
    from gdmix_amd.solver import NUM_CLASSES
    sync = torch.cuda.synchronize
    parts = np.unique(share.pid)
    nr = int(gather_rows([min(rounds, parts.size)])[:, 0].min())
    comm = _Comm(None, solver.device)
    model = SizeCostModel()
    totals = np.zeros((2, SizeCostModel.BUCKETS))
    out = []
    for k in range(nr):
        idx = np.flatnonzero(share.pid == parts[k])
        raw = _slice_entities(share.raw_dev, int(idx[0]), int(idx[-1]) + 1)
        n_k, z_k = share.n[idx], share.z[idx]
        solver.pack(raw); sync()      # (first touch of this partition's pages: not part of either timing)
        dist.barrier()
        t0 = time.perf_counter()
        pk = solver.pack(raw)
        rs = solver.solve(pk, opts)
        sync()
        plain = (time.perf_counter() - t0) * 1e3
        dist.barrier()
        plain_job = (time.perf_counter() - t0) * 1e3
        class_ms = np.array(solver.last_solve_ms())
        cls = _entity_classes(pk, torch)
        theta_plain = rs.theta_thr.clone()
        row = {"round": k, "partition": int(parts[k]), "entities": int(idx.size), "nnz": int(z_k.sum())}
        reb = None
        if rebalance:
            wire = _wire_from_raw(raw)
            sync(); dist.barrier()
            t0 = time.perf_counter()
            rb = Rebalancer(n_k, z_k, wire, cost=model.cost(z_k), order=model.order(z_k), comm=comm, tolerance=tolerance)
            b0 = comm.bytes_sent
            work = rb.exchange()
            pk2 = solver.pack(solver.widen(work))
            rs2 = solver.solve(pk2, opts)
            fp = pk2.ent_feat_ptr()
            ints = torch.stack([rs2.nit, rs2.nfev, rs2.status], dim=1)
            flts = torch.stack([rs2.fval, rs2.gnorm], dim=1)
            _, th, _, _, _, _ = rb.give_back((fp[1:] - fp[:-1]) + 1, rs2.theta_thr, None, pk2.unique_global(), ints, flts)
            sync()
            mine = (time.perf_counter() - t0) * 1e3
            dist.barrier()
            job = (time.perf_counter() - t0) * 1e3
            scale = torch.clamp(theta_plain.abs(), min=1e-3)
            diff = float(((th - theta_plain).abs() / scale).max().item()) if th.numel() else 0.0
            reb = [mine, job, comm.bytes_sent - b0, sum(int(x.size) for x in rb.sent), diff]
            del rb, work, pk2, rs2, th
        # what this round measured prices the next one (summed over ranks)
        tt = torch.from_numpy(SizeCostModel.totals(cls, z_k, class_ms, NUM_CLASSES)).to(coll_dev)
        dist.all_reduce(tt)
        totals += tt.cpu().numpy()
        model = SizeCostModel.from_totals(totals)
        rows = gather_rows([plain, plain_job] + (reb or [0, 0, 0, 0, 0]))
        row.update(plain_ms=[round(float(x), 2) for x in rows[:, 0]], plain_round_ms=float(rows[:, 1].max()),
                   imbalance=float(rows[:, 0].max() / rows[:, 0].mean()))
        if rebalance:
            row.update(rebalanced_ms=[round(float(x), 2) for x in rows[:, 2]], rebalanced_round_ms=float(rows[:, 3].max()),
                       bytes_sent=[int(x) for x in rows[:, 4]], entities_sent=[int(x) for x in rows[:, 5]],
                       max_rel_diff_vs_plain=float(rows[:, 6].max()), priced_by="non-zeros" if k == 0 else f"measured costs of rounds 0..{k - 1}")
        out.append(row)
        del pk, rs, raw
    res = {"what": "one partition per worker and round, workers in lockstep: plain pack + solve, then the same round through Rebalancer.exchange / "
                   "give_back priced by what the rounds before it measured (rebalance.SizeCostModel)", "rounds": out,
           "plain_ms_total": float(sum(r["plain_round_ms"] for r in out))}
    if rebalance and out:
        res["rebalanced_ms_total"] = float(sum(r["rebalanced_round_ms"] for r in out))
    return res


def _slice_entities(raw, e0, e1):
    """Entities [e0, e1) of a raw batch in HBM as a raw batch of their own (views; the pointers rebased)."""
    erp = raw["ent_row_ptr"]
    r0, r1 = int(erp[e0].item()), int(erp[e1].item())
    rnp = raw["row_nnz_ptr"]
    z0, z1 = int(rnp[r0].item()), int(rnp[r1].item())
    return dict(E=e1 - e0, N=r1 - r0, Z=z1 - z0, ent_row_ptr=erp[e0:e1 + 1] - r0, row_nnz_ptr=rnp[r0:r1 + 1] - z0,
                col_global=raw["col_global"][z0:z1], val=raw["val"][z0:z1], y=raw["y"][r0:r1], offset=raw["offset"][r0:r1],
                weight=None if raw["weight"] is None else raw["weight"][r0:r1])


def partition_rounds(share, solver, opts, rounds):
    """The product path's granularity: worker r trains ONE partition per round (drivers/random_effect_driver.py:65-68 walks
    partitions[r::R]). -> per round the milliseconds (pack + solve, best of two) and non-zeros of this share's partition of that round."""
    import torch
    parts = np.unique(share.pid)[:rounds]
    out = []
    for K in parts:
        idx = np.flatnonzero(share.pid == K)
        raw = _slice_entities(share.raw_dev, int(idx[0]), int(idx[-1]) + 1)
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pk = solver.pack(raw)
            solver.solve(pk, opts)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        out.append({"partition": int(K), "entities": int(idx.size), "nnz": int(share.z[idx].sum()), "largest_nnz": int(share.z[idx].max()), "ms": best})
    return out


def projection_leg(name, ranks, solver, opts, c5_entities, steps=2, warmup=1, ml_entities=None, tolerance=0.05, rounds=4):
    """One GPU: the `ranks` shares of population `name` one after another (see the module docstring)."""
    import torch
    from gdmix_amd.rebalance import CostModel, choose_entities, plan_transfers
    from gdmix_amd.solver import NUM_CLASSES
    solver.set_timing(True)
    per, totals, kept = [], np.zeros((2, NUM_CLASSES)), []
    for r in range(ranks):
        share = make_share(name, ranks, r, solver, c5_entities, ml_entities)
        packed, res, own_s, _, class_ms = _timed_steps(solver, share, opts, steps, warmup, torch.cuda.synchronize, lambda: None)
        cls = _entity_classes(packed, torch)
        totals += CostModel.totals(cls, share.z, class_ms, NUM_CLASSES)
        kept.append((cls, share.z, share.n))
        names = [n for n, _ in solver.class_counts(packed)]
        cnts = np.bincount(cls, minlength=NUM_CLASSES)
        top = sorted(((float(class_ms[c]), names[c], int(cnts[c])) for c in range(NUM_CLASSES) if class_ms[c] > 0), reverse=True)[:5]
        per.append({"rank": r, "largest_launches": [{"kernel": k, "entities": e, "ms": round(ms, 3)} for ms, k, e in top], "entities": share.E, "samples": share.N, "nnz": share.Z, "ms_per_step": float(np.median(share.step_ms)), "ms_per_step_mean": own_s / steps * 1e3,
                    "pack_ms": share.pack_ms, "solve_kernel_ms": float(class_ms.sum()), "converged": _converged(res),
                    "largest_nnz": int(share.z.max()) if share.E else 0, "step_wall_ms": list(share.step_ms),
                    "generate_s": round(share.gen_s, 2)})
        total_entities, partitions = share.total_entities, share.partitions
        if name == "c5" and rounds > 0:
            per[-1]["rounds"] = partition_rounds(share, solver, opts, rounds)
        del share, packed, res
        torch.cuda.empty_cache()
    ms = np.array([p["ms_per_step"] for p in per])
    conv = sum(p["converged"] for p in per)
    out = {"workload": name, "what": STRONG_WORKLOADS[name][0], "ranks": ranks, "partitions": partitions,
           "projection": f"the {ranks} shares were solved one after another on ONE MI355X (no collective on the data path): ms = the slowest share; a "
                         "share's ms_per_step = the MEDIAN of its timed steps (step_wall_ms lists them, ms_per_step_mean is their mean: one host hiccup - "
                         "19.6 ms in a 2 ms step, once in a run - would otherwise be the slowest rank of the job)",
           "total_entities": sum(p["entities"] for p in per), "total_nnz": sum(p["nnz"] for p in per),
           "ms": float(ms.max()), "entities_per_s": conv / (float(ms.max()) * 1e-3), "imbalance": float(ms.max() / ms.mean()),
           "ms_mean": float(max(p["ms_per_step_mean"] for p in per)),      # the slowest share by the MEAN of its timed steps (hiccups included)
           "entities_per_s_by_mean": conv / (float(max(p["ms_per_step_mean"] for p in per)) * 1e-3),
           "sum_of_shares_ms": float(ms.sum()), "per_rank": per}
    assert out["total_entities"] == total_entities
    if all("rounds" in p for p in per):
        # one partition per worker and round, all workers in lockstep (what --rebalance_entities balances): a round lasts as long as its
        # slowest partition
        rr = []
        for k in range(min(len(p["rounds"]) for p in per)):
            ms_k = np.array([p["rounds"][k]["ms"] for p in per])
            big = max(p["rounds"][k]["largest_nnz"] for p in per)
            rr.append({"round": k, "ms": [round(float(x), 2) for x in ms_k], "imbalance": float(ms_k.max() / ms_k.mean()), "largest_entity_nnz": int(big)})
        out["partition_rounds"] = {"what": "the product path's granularity: one partition per worker and round, workers in lockstep; ms = pack + solve of "
                                           "that partition alone on this device (best of three)", "rounds": rr,
                                   "mean_imbalance": float(np.mean([r["imbalance"] for r in rr])), "worst_imbalance": float(max(r["imbalance"] for r in rr))}
        for p in per:
            del p["rounds"]
    # the re-balancing plan on measured costs (not executed here: the exchange needs the other ranks)
    additive = np.arange(NUM_CLASSES) <= NUM_CLASSES - 5
    model = CostModel.from_totals(totals, additive)
    costs = [model.cost(c, z) for c, z, _ in kept]
    loads = np.array([c.sum() for c in costs])
    T = plan_transfers(loads, tolerance)
    after = loads.copy()
    bytes_moved = 0.0
    ents_moved = 0
    for i in range(ranks):
        if T[i].sum() <= 0:
            continue
        cls, z, n = kept[i]
        sent = choose_entities(costs[i], T[i], model.order(cls, z))
        for j, ix in enumerate(sent):
            after[i] -= costs[i][ix].sum()
            after[j] += costs[i][ix].sum()
            # wire form: int32 feature id + fp32 value per non-zero; count, label, offset per sample; a count per entity
            bytes_moved += float(z[ix].sum()) * 8.0 + float(n[ix].sum()) * 12.0 + float(ix.size) * 4.0
            ents_moved += int(ix.size)
    out["rebalance_plan"] = {"predicted_load_ms": [round(float(x), 3) for x in loads], "after_ms": [round(float(x), 3) for x in after],
                             "predicted_imbalance": float(loads.max() / loads.mean()), "after_imbalance": float(after.max() / after.mean()),
                             "entities_to_move": ents_moved, "wire_bytes_to_move": bytes_moved, "tolerance": tolerance,
                             "note": "plan only (rebalance.plan_transfers / choose_entities on rebalance.CostModel costs measured here); executed by "
                                     "strong_leg when the job has ranks"}
    return out


def c5_full_share_leg(solver, opts, device_index, total_entities=100_000_000, ranks=8, partitions=1024, projected_rounds=16, tolerance=0.05,
                      contexts=3):
    """BASELINE configs[4] at its real per-GPU share and at the product path's granularity: ONE population of
    `total_entities` Zipf-sized entities hashed into `partitions` partitions; worker 0 of `ranks` owns
    partitions[0::ranks] = 128 of 1 024 = 12.5 M entities and trains them ONE PARTITION PER ROUND
    (drivers/random_effect_driver.py:60-68). Every partition of the share is generated in HBM (its own Philox stream), then
      serial     : the 128 rounds one after another on one context — pack + solve per round, a round's ms
      pipelined  : the same 128 partitions over `contexts` contexts / streams round robin (how model.py's host pipeline and
                   bench.py's hand-over leg keep the device busy across the rounds' tails) — the share's wall time
    and, for the first `projected_rounds` rounds, ALL `ranks` workers' partitions of the round solved one after another on this
    device (a projection: the data path has no collective) plain and with the re-balancer's plan applied — rebalance.plan_transfers /
    choose_entities on rebalance.SizeCostModel costs measured by the rounds before (the first round: non-zeros), the travelling
    entities really moved between the rounds' batches in HBM (synthetic.subset_raw / concat_raw) and the new batches timed; the
    exchange itself (wire bytes over xGMI) cannot run on one GPU and is reported as bytes."""
    import threading
    import torch
    from gdmix_amd import synthetic
    from gdmix_amd.rebalance import SizeCostModel, choose_entities, plan_transfers
    from gdmix_amd.solver import NUM_CLASSES, REDeviceSolver
    sync = torch.cuda.synchronize
    dev_pids = lambda ids, parts: solver.partition_ids(np.ascontiguousarray(ids, np.int64), parts).cpu().numpy()
    t0 = time.perf_counter()
    pop = synthetic.C5Population(total_entities, partitions, partition_ids_fn=dev_pids)
    t_pop = time.perf_counter() - t0
    mine = synthetic.rank_partitions(partitions, ranks, 0)
    t0 = time.perf_counter()
    share = [pop.partition(K, solver.device) for K in mine]
    sync()
    t_gen = time.perf_counter() - t0
    E = sum(r["E"] for r, _, _ in share)
    Z = sum(r["Z"] for r, _, _ in share)
    solver.set_timing(True)

    def step(s, raw):
        pk = s.pack(raw)
        rs = s.solve(pk, opts)
        return pk, rs

    def converged(rs):
        st = rs.status
        return int(((st >= 0) & (st <= 2)).sum().item())
    for raw, _, _ in share[:2]:
        step(solver, raw)
    sync()
    # ---- serial: one round after another
    round_ms, conv = [], 0
    t_serial = time.perf_counter()
    for raw, _, _ in share:
        t = time.perf_counter()
        pk, rs = step(solver, raw)
        sync()
        round_ms.append((time.perf_counter() - t) * 1e3)
        conv += converged(rs)
        del pk, rs
    t_serial = time.perf_counter() - t_serial
    # ---- pipelined over `contexts` contexts
    ws = [REDeviceSolver(device_index) for _ in range(contexts)]
    streams = [torch.cuda.Stream(device=solver.device) for _ in range(contexts)]
    for i, s in enumerate(ws):
        with torch.cuda.stream(streams[i]):
            step(s, share[i][0])
            streams[i].synchronize()
    conv_p = [0] * contexts
    start = threading.Barrier(contexts + 1)

    def run(i):
        start.wait()
        with torch.cuda.stream(streams[i]):
            for k in range(i, len(share), contexts):
                pk, rs = step(ws[i], share[k][0])
                conv_p[i] += converged(rs)      # (the read-back of the statuses also ends the round on this stream)
                del pk, rs
    threads = [threading.Thread(target=run, args=(i,)) for i in range(contexts)]
    for th in threads:
        th.start()
    sync()
    start.wait()
    t_pipe = time.perf_counter()
    for th in threads:
        th.join()
    sync()
    t_pipe = time.perf_counter() - t_pipe
    for s in ws:
        s.close()
    # ---- what several partitions per device batch would buy (NOT what the product path does: it trains one partition per round, as the
    # reference's driver does): K consecutive partitions of the share concatenated in HBM (untimed), then pack + solve of the batch
    batched = {}
    for K in (2, 4, 8):
        t_k, conv_k = 0.0, 0
        for g0 in range(0, len(share), K):
            raw = synthetic.concat_raw([r for r, _, _ in share[g0:g0 + K]])
            sync()
            t = time.perf_counter()
            pk, rs = step(solver, raw)
            sync()
            t_k += time.perf_counter() - t
            conv_k += converged(rs)
            del pk, rs, raw
        batched[str(K)] = {"s": t_k, "entities_per_s": E / t_k, "converged": conv_k}
        torch.cuda.empty_cache()
    rm = np.array(round_ms)
    out = {"what": f"worker 0 of {ranks}: its {len(mine)} partitions of ONE population of {total_entities} Zipf-sized entities in {partitions} Java-hashed partitions "
                   "(SURVEY 8(d)), one partition per round, resident in HBM; s = the share's wall time over three contexts, serial_s = one context",
           "entities": int(E), "nnz": int(Z), "partitions": len(mine), "converged": int(conv), "converged_pipelined": int(sum(conv_p)),
           "s": t_pipe, "entities_per_s": E / t_pipe, "serial_s": t_serial, "serial_entities_per_s": E / t_serial, "contexts": contexts,
           "round_ms_p50": float(np.percentile(rm, 50)), "round_ms_p99": float(np.percentile(rm, 99)), "round_ms_max": float(rm.max()),
           "round_ms_mean": float(rm.mean()), "round_ms": [round(float(x), 2) for x in rm],
           "largest_entity_nnz": int(max(int(n.max()) for _, n, _ in share) * pop.k),
           "population_s": round(t_pop, 1), "generate_s": round(t_gen, 1),
           "partitions_per_batch": {"what": "the same share with K consecutive partitions concatenated into one device batch (one context): what the driver "
                                            "gets on the device with GDMIX_PARTITIONS_PER_BATCH = K (model.plan_group; default 1: one partition per round)", **batched}}
    # ---- the first rounds of the whole job: every worker's partition of the round, plain and with the plan applied
    model, totals = SizeCostModel(), np.zeros((2, SizeCostModel.BUCKETS))
    rounds = []

    def timed(raw):
        best, keep = None, None
        for _ in range(2):       # best of two (the first touches the batch's pages)
            sync()
            t = time.perf_counter()
            pk, rs = step(solver, raw)
            sync()
            dt = (time.perf_counter() - t) * 1e3
            if best is None or dt < best:
                best = dt
            keep = (pk, rs)
        return best, keep
    for k in range(min(projected_rounds, len(mine))):
        batches = [share[k] if r == 0 else pop.partition(synthetic.rank_partitions(partitions, ranks, r)[k], solver.device) for r in range(ranks)]
        plain, cls_ms = [], []
        for raw, n, _ in batches:
            ms, (pk, rs) = timed(raw)
            plain.append(ms)
            cls_ms.append((_entity_classes(pk, torch), np.array(solver.last_solve_ms())))
            del pk, rs
        z = [n * pop.k for _, n, _ in batches]
        costs = [model.cost(zz) for zz in z]
        loads = np.array([c.sum() for c in costs])
        T = plan_transfers(loads, tolerance)
        sent = [choose_entities(costs[i], T[i], model.order(z[i])) if T[i].sum() > 0 else [np.zeros(0, np.int64)] * ranks for i in range(ranks)]
        moved = sum(int(ix.size) for s_ in sent for ix in s_)
        bytes_moved = sum(float(z[i][ix].sum()) * 8.0 + float(batches[i][1][ix].sum()) * 12.0 + ix.size * 4.0 for i in range(ranks) for ix in sent[i])
        after = list(plain)
        if moved:
            after = []
            for j in range(ranks):
                gone = np.concatenate(sent[j]) if T[j].sum() > 0 else np.zeros(0, np.int64)
                keep_ix = np.setdiff1d(np.arange(batches[j][0]["E"]), gone)
                parts_j = [synthetic.subset_raw(batches[j][0], batches[j][1], keep_ix)[0] if gone.size else batches[j][0]]
                for i in range(ranks):
                    if i != j and sent[i][j].size:
                        parts_j.append(synthetic.subset_raw(batches[i][0], batches[i][1], sent[i][j])[0])
                ms, (pk, rs) = timed(synthetic.concat_raw(parts_j))
                after.append(ms)
                del pk, rs, parts_j
        for (cls, cms), zz in zip(cls_ms, z):       # what this round measured prices the next one
            totals += SizeCostModel.totals(cls, zz, cms, NUM_CLASSES)
        model = SizeCostModel.from_totals(totals)
        pl, af = np.array(plain), np.array(after)
        rounds.append({"round": k, "plain_ms": [round(float(x), 2) for x in pl], "plain_round_ms": float(pl.max()), "imbalance": float(pl.max() / pl.mean()),
                       "rebalanced_ms": [round(float(x), 2) for x in af], "rebalanced_round_ms": float(af.max()),
                       "imbalance_after": float(af.max() / af.mean()), "entities_moved": int(moved), "wire_bytes_moved": float(bytes_moved),
                       "priced_by": "non-zeros" if k == 0 else f"measured costs of rounds 0..{k - 1}"})
        del batches
        torch.cuda.empty_cache()
    if rounds:
        pr = np.array([r["plain_round_ms"] for r in rounds])
        rr = np.array([r["rebalanced_round_ms"] for r in rounds])
        mean_round = np.array([np.mean(r["plain_ms"]) for r in rounds])
        out["imbalance_vs_8_rank_mean_round"] = float(np.mean([r["plain_ms"][0] for r in rounds]) / mean_round.mean())
        out["projected_rounds"] = {
            "what": f"rounds 0..{len(rounds) - 1} of the whole {ranks}-worker job, every worker's partition of the round solved one after another on this device "
                    "(projection), plain and with the re-balancer's plan applied to the batches in HBM (exchange itself not timed: bytes reported)",
            "rounds": rounds, "plain_ms_total": float(pr.sum()), "rebalanced_ms_total": float(rr.sum()),
            "mean_imbalance": float(np.mean([r["imbalance"] for r in rounds])), "mean_imbalance_after": float(np.mean([r["imbalance_after"] for r in rounds])),
            "exchange_ms_estimate_per_round": float(np.mean([r["wire_bytes_moved"] for r in rounds]) / 153e9 * 1e3),
            "exchange_estimate_note": "mean wire bytes of a round / one xGMI link (153 GB/s); a direct all-to-all uses up to 7 links per GPU"}
    return out
