#!/usr/bin/env python3
"""bench.py — random-effect entities converged / second on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the device hot path over one resident batch: gdmix_re_pack (per-entity
unique/local-index/CSC build) followed by gdmix_re_solve (the whole per-entity L-BFGS on the device),
on the configuration BASELINE.json's metric is quoted on for one GPU: configs[1], synthetic 1M
entities x avg 64 nnz (n ~ max(1, Poisson(16)), k = 4 distinct uniform columns of D = 1024, seed 20240601: the generator of
SURVEY.md §8(d), gdmix_amd.synthetic.make_survey_batch), solver options of the
shipped MovieLens config (l2 = 1, regularize_bias = false, m = 10, max_iter = 100, tol = 1e-12).
Inputs are resident in HBM before the timed region; outputs stay in HBM. With N > 1 every rank owns
its own shard of 1M entities (weak scaling; entities are independent, no data-path collective); started
without WORLD_SIZE in the environment, `--gpus N` re-executes the script under torch.distributed.run with N
ranks, one device each (spawn_ranks), and a WORLD_SIZE that differs from --gpus is an error.

`--workload` picks another BASELINE configuration for the step: ml20m_user / ml20m_movie (C3: MovieLens-20M
entity sizes), c5share (C5: one GPU's share, 4M Zipf entities generated on the device), zipf, ...; the
default C2 run also measures ml20m_user, ml20m_movie and c5share briefly and reports them under
detail.workloads (--no-other-workloads skips that; it is never part of `value`).

ONE population over the ranks (BASELINE configs[2] and [4]; bench_strong.py): with --gpus N > 1 the default run also measures three
populations split the reference's way — entity -> partition by the Java hash of its decimal id, partition -> rank by partitions[rank::N] —
plain and through the re-balancer (top-level `strong_scaling`; `--scaling strong --workload c5|ml20m_user|ml20m_movie` makes one of them
the headline); with --gpus 1 it solves the shares of an 8-rank job one after another on this device (detail.strong_projection: the data
path has no collective, so max over shares is the job time apart from barriers — labelled a projection), including the first
per-partition rounds of the C5 job, the granularity at which the product path re-balances.

One JSON line is printed by rank 0. `roofline` is the HBM roofline of the dominant kernel
(the size-class launch that takes the largest share of a step) by the algorithmic-bytes formula of SURVEY.md §8(d); `cpu_baseline` times the
CPU oracle (oracle/re_oracle.c, a port) on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # a process setting, made by the entry point before the first device call (gdmix_amd/gdmix.py: process_defaults)

SPREAD_DEFAULT = int(os.environ.get("GDMIX_RE_SPREAD", "4"))   # queues the large size classes are dealt over (gdmix_re_set_spread)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--entities", type=int, default=1_000_000, help="entities per GPU (C2: 1M)")
    ap.add_argument("--mean-n", type=int, default=16)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--cpu-sample", type=int, default=0, help="entities per pass of the CPU baseline (0 = 200k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive hand-over measurement")
    ap.add_argument("--no-alone", action="store_true",
                    help="skip roofline.alone (four more steps with the size classes one after another): for a run under rocprofv3, whose "
                         "per-kernel averages should hold the timed schedule only")
    ap.add_argument("--no-fe", action="store_true", help="skip the fixed-effect evaluation leg (detail.fixed_effect_eval)")
    ap.add_argument("--no-cli", action="store_true", help="skip the end-to-end leg through the CLI (detail.cli_end_to_end)")
    ap.add_argument("--cli-entities", type=int, default=1_000_000, help="entities of the end-to-end leg (partitions of 125 k)")
    ap.add_argument("--cli-c5-entities", type=int, default=200_000, help="entities of the C5-shaped end-to-end leg (detail.cli_end_to_end_c5)")
    ap.add_argument("--fe-rows", type=int, default=4_000_000, help="samples of the fixed-effect leg's shard (x 32 non-zeros, 100k features)")
    ap.add_argument("--giant-nnz", type=int, default=-1, help="override the device-wide kernel threshold (exploration)")
    ap.add_argument("--team-nnz", type=int, default=-1, help="override the lowest team-tier threshold (exploration)")
    ap.add_argument("--tall-min-n", type=int, default=-1, help="override the tall kernel's sample threshold (exploration)")
    ap.add_argument("--kernel-mask", type=int, default=-1, help="gdmix_re_set_kernel_mask (exploration)")
    ap.add_argument("--tall-split-n", type=int, default=-1, help="override the tall kernel's one-CU-per-entity threshold (exploration)")
    ap.add_argument("--lbfgs-m", type=int, default=10, help="history pairs (exploration: what the direction step costs; the shipped config has 10)")
    ap.add_argument("--solve-only", action="store_true", help="time gdmix_re_solve alone (batch packed once)")
    ap.add_argument("--ranks-share-device", action="store_true",
                    help="test hook for a 1-GPU box: every rank uses cuda:0 and the collectives go over gloo (numbers are meaningless)")
    ap.add_argument("--workload", default="c2", choices=list(WORKLOADS),
                    help="c2 (default, the configuration BASELINE.json's metric is quoted on for one GPU) or another of BASELINE's "
                         "configurations at its own entity sizes: " + "; ".join(f"{k} = {v}" for k, v in WORKLOADS.items()))
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip detail.workloads: C3 (MovieLens-20M per-user / per-movie) and C5's per-GPU share, two steps each, next to the C2 headline")
    ap.add_argument("--c5-entities", type=int, default=4_000_000, help="entities of the c5share workload per GPU (C5's share is 12.5 M)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank owns its own batch, the headline of the bench contract. strong: ONE population split over "
                         "the ranks by the reference's partition hash and partitions[rank::ranks] (bench_strong.py); --workload picks "
                         "ml20m_user / ml20m_movie / c5 (default c5), `value` is that job's entities/s")
    ap.add_argument("--no-strong", action="store_true",
                    help="with --gpus N > 1 the default run also measures the three strong-scaling populations (top-level `strong_scaling`); skip that")
    ap.add_argument("--no-rebalance", action="store_true", help="strong-scaling legs: skip the run through the re-balancer")
    ap.add_argument("--rebalance-tolerance", type=float, default=0.02,
                    help="strong-scaling legs: ranks within (1 + tolerance) x the mean cost neither send nor receive (the product's default is 0.05)")
    ap.add_argument("--strong-steps", type=int, default=4, help="timed steps of each strong-scaling / projection leg (after two warm-up steps)")
    ap.add_argument("--project-ranks", type=int, default=8,
                    help="--gpus 1, default workload: also solve the shares of an N-rank job one after another on this GPU "
                         "(detail.strong_projection; 0 = skip)")
    ap.add_argument("--c5-full-entities", type=int, default=100_000_000,
                    help="--gpus 1, default workload: BASELINE configs[4] at its real per-GPU share — worker 0's 128 partitions of ONE population of this "
                         "many Zipf-sized entities in 1 024 partitions, one partition per round (detail.c5_full_share; 0 = skip)")
    ap.add_argument("--c5-full-rounds", type=int, default=16, help="rounds of the whole 8-worker job projected on this device, plain and with the re-balancing plan applied")
    ap.add_argument("--detail-file", default="", help="where the full result goes (default gpurun_out/bench_detail.json); the stdout line is the short form")
    ap.add_argument("--print-detail", action="store_true", help="also print the full result as one JSON line on stderr")
    ap.add_argument("--ml-entities", type=int, default=0, help="strong legs: keep this many MovieLens entities (tests at reduced size; 0 = all)")
    return ap.parse_args()


WORKLOADS = {
    "c5": "strong scaling only: ONE Zipf population of --c5-entities x ranks entities hashed into 1 024 partitions (bench_strong.py)",
    "c2": "C2, synthetic 1 M entities x avg 64 nnz (SURVEY 8(d) generator)",
    "c5mean": "C5's mean shape without the tail (32 samples x 8 nnz, D = 65 536), stratified generator",
    "zipf": "the same with Zipf sizes (round-1/2 exploration shape, tail not capped at 2^20 nnz)",
    "ml_user": "MovieLens per-user entities at ML-100K sizes (C1 shape)",
    "ml_movie": "MovieLens per-movie entities at ML-100K sizes (C1 shape)",
    "ml20m_user": "C3: MovieLens-20M per-user entities (138 493 users, 16 M training rows, n up to 7.4 k, p <= 21)",
    "ml20m_movie": "C3: MovieLens-20M per-movie entities (26 744 movies, 16 M training rows, head of 54 k samples, p <= 25)",
    "c5share": "C5's per-GPU share: --c5-entities Zipf-sized entities (P(nnz >= x) ~ x^-1.2 on [8, 2^20], mean 256, D = 65 536), generated in HBM",
}


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run (one process
    per GPU, RCCL) and hand its exit code back. With --ranks-share-device (test hook for a 1-GPU box) the ranks share
    cuda:0 and talk over gloo."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if not a.ranks_share_device and have < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} HIP device(s) visible (one process per GPU; "
                         "--ranks-share-device runs the ranks on cuda:0 over gloo as a harness test)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


HOST_LEG_REPEATS = 3      # the host-bound legs (CLI, hand-over, chain) are repeated: one run of them on a shared box says little


def _spread(xs):
    """Median, extremes and every run of a repeated host-bound measurement (a +- 20 % box-to-box spread makes a
    single run of these legs meaningless for a round-over-round comparison)."""
    xs = [float(x) for x in xs]
    return {"median": float(np.median(xs)), "min": min(xs), "max": max(xs), "runs": [round(x, 6) for x in xs]}


def cpu_baseline(sub, opts_kw, min_seconds=10.0):
    """Time the fp64 CPU restatement (the oracle, kind 'port') on the entities of `sub`, repeated until at least
    `min_seconds` of wall time, one oracle call per host core (ctypes releases the GIL), entities dealt to the cores in
    contiguous runs of about equal non-zero count."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    cores = os.cpu_count() or 1
    pk = oracle.pack(sub.ent_row_ptr, sub.row_nnz_ptr, sub.col_global)
    o = oracle.make_opts(**opts_kw)
    E = sub.E
    work = np.concatenate([[0], np.cumsum(sub.ent_nnz() + 64)])
    bounds = np.searchsorted(work, np.linspace(0, work[-1], cores + 1)).astype(int)
    bounds[0], bounds[-1] = 0, E

    def run(i):
        r = oracle.solve(pk, sub.val, sub.y, sub.offset, sub.weight, o, e_begin=int(bounds[i]), e_end=int(bounds[i + 1]))
        return int(np.isin(r["status"][bounds[i]:bounds[i + 1]], (0, 1, 2)).sum())
    conv, passes = 0, 0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        while True:
            conv += sum(ex.map(run, range(cores)))
            passes += 1
            dt = time.perf_counter() - t0
            if dt >= min_seconds or passes >= 200:
                break
    return conv / dt, cores, E, dt, passes


def host_handover(batch, opts, device_index, P, partitions=24, workers=3):
    """Partitions handed over from host memory, pipelined: `workers` threads, each with its own context, HIP stream and
    page-locked staging, take partitions round robin — upload k+1 || widen + pack + solve k || download k-1. The batch
    crosses PCIe in the 32-bit wire form (gdmix_re_wire_batch: counts, int32 feature ids, byte labels); the thresholded
    coefficients come back as float64. Every partition is the same C2 batch (its wire arrays sit in page-locked memory, where
    the native reader would have decoded them). Also timed: one partition alone on one stream (no overlap)."""
    import threading
    import torch
    from gdmix_amd.solver import REDeviceSolver
    wire = batch.to_wire()
    keys = [k for k in REDeviceSolver.WIRE_ARRAYS if wire[k] is not None]
    h2d = sum(wire[k].nbytes for k in keys)

    class Worker:
        def __init__(self):
            self.solver = REDeviceSolver(device_index)
            self.stream = torch.cuda.Stream(device=self.solver.device)
            self.wire = dict(wire)
            for k in keys:   # this worker's page-locked copy of the partition
                self.wire[k] = torch.from_numpy(wire[k]).pin_memory()
            self.theta_host = torch.empty(P, dtype=torch.float64).pin_memory()
            self.status_host = torch.empty(batch.E, dtype=torch.int32).pin_memory()
            self.index_host = torch.empty(P, dtype=torch.int32).pin_memory()
            self.converged = 0

        def one(self, with_index=False):
            s = self.solver
            with torch.cuda.stream(self.stream):
                rd = s.widen(s.upload_wire(self.wire))
                pk = s.pack(rd)
                res = s.solve(pk, opts)
                self.theta_host[:pk.P].copy_(res.theta_thr, non_blocking=True)
                self.status_host.copy_(res.status, non_blocking=True)
                if with_index:   # local -> global feature index of every coefficient (np.unique's first output), as int32
                    self.index_host[:pk.D].copy_(pk.unique_global().to(torch.int32), non_blocking=True)
                self.stream.synchronize()
            st = self.status_host.numpy()
            self.converged += int(((st >= 0) & (st <= 2)).sum())

    ws = [Worker() for _ in range(workers)]
    for w in ws:
        w.one()          # warm-up: allocator, kernels
        w.converged = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ws[0].one()
    serial = time.perf_counter() - t0
    ws[0].converged = 0
    def pipelined(with_index):
        start = threading.Barrier(workers + 1)

        def run(i):
            start.wait()
            for _ in range(i, partitions, workers):
                ws[i].one(with_index)
        threads = [threading.Thread(target=run, args=(i,)) for i in range(workers)]
        for w in ws:
            w.converged = 0
        for th in threads:
            th.start()
        start.wait()
        t0 = time.perf_counter()
        for th in threads:
            th.join()
        dt = time.perf_counter() - t0
        return dt, sum(w.converged for w in ws)
    # one run to settle (thread start-up, the first touch of the staging blocks), then HOST_LEG_REPEATS: the median is the figure
    pipelined(False)
    runs = sorted(pipelined(False) for _ in range(HOST_LEG_REPEATS))
    dt, conv = runs[len(runs) // 2]
    runs_ix = sorted(pipelined(True) for _ in range(HOST_LEG_REPEATS))
    dt_ix, conv_ix = runs_ix[len(runs_ix) // 2]
    for w in ws:
        w.solver.close()
    return {"ms_per_partition": dt / partitions * 1e3, "entities_per_s": conv / dt, "partitions": partitions, "streams": workers,
            "entities_per_s_spread": _spread(c / t for t, c in runs),
            "h2d_bytes_per_partition": h2d, "d2h_bytes_per_partition": P * 8 + batch.E * 4,
            "with_feature_index": {"ms_per_partition": dt_ix / partitions * 1e3, "entities_per_s": conv_ix / dt_ix,
                                   "d2h_bytes_per_partition": P * 8 + batch.E * 4 + (P - batch.E) * 4,
                                   "what": "the same plus the local -> global feature index of every coefficient (int32), which a model file needs"},
            "serial_one_stream": {"ms": serial * 1e3, "entities_per_s": batch.E / serial},
            "what": "page-locked 32-bit wire batch -> H2D -> gdmix_re_widen -> pack -> solve -> D2H thresholded theta (f64) + status, "
                    f"{workers} streams, partitions round robin; serial_one_stream = the same for one partition without overlap"}


def fixed_effect_leg(solver, rows, nnz_per_row=32, features=100_000, iters=20, dist="uniform"):
    """SURVEY.md 8(f) N1 next to the headline: one worker's shard (rows x 32 uniform columns of 100k features, logistic, m = 10)
    resident in HBM, L-BFGS by the stepping kernels of include/gdmix_fe.h; time per evaluation (two passes over the non-zeros
    + the replicated step) and the algorithmic bytes it moves (tools/fe_bench.py has the same accounting)."""
    import torch
    from gdmix_amd import fixed_effect as fe
    from gdmix_amd.solver import SolverOptions
    rng = np.random.default_rng(0)
    n, k, D = rows, nnz_per_row, features
    if dist == "zipf":   # feature j with probability ~ 1/(j+1) (tools/fe_bench.py): the most frequent one holds 1/ln(D) of all entries
        cols = np.minimum((float(D + 1) ** rng.random(n * k)).astype(np.int64) - 1, D - 1)
    else:
        cols = rng.integers(0, D, n * k, dtype=np.int64)
    vals = (rng.random(n * k, dtype=np.float32) - 0.5) * 2.0
    y = (rng.random(n, dtype=np.float32) < 0.5).astype(np.float32)
    off = np.zeros(n, np.float32)
    rp = np.arange(n + 1, dtype=np.int64) * k
    batch, _ = fe.shard_as_batch(rp, cols, vals, y, off, None, True)
    packed = solver.pack(batch)
    opts = SolverOptions(l2=1.0, regularize_bias=True, has_intercept=True, m=10, max_iter=iters, threshold=0.0, sum_loss=True)
    out = None
    for _ in range(2):   # the second fit is the measured one
        prob = fe._SteppingProblem(solver, packed, D, opts, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fe.run_stepping_loop(prob)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        _, info = prob.result()
        rows_ms, cols_ms = prob.last_eval_ms()
        rows_b, cols_b = prob.stream_bytes()
        prob.close()
        nfev = int(info["nfev"])
        Z, P, m = n * k, D + 1, 10
        alg = 16.0 * Z + 32.0 * n + (4 + 2 * m) * 8.0 * P
        ms = dt * 1e3 / nfev
        out = {"ms_per_evaluation": ms, "rows_pass_ms": rows_ms, "cols_pass_ms": cols_ms, "evaluations": nfev, "alg_bytes_per_evaluation": alg,
               "GBps": alg / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "entry_bytes_streamed": {"rows_pass": rows_b, "cols_pass": cols_b, "per_nnz": (rows_b + cols_b) / float(Z),
                                        "note": "what the passes' own copies hold: 8 B per entry, units of the column pass in the 6-byte form of round 5 "
                                                "(fp32 value + 16-bit {key delta, accumulator}); the roofline figure stays on the algorithmic 16 B per non-zero"},
               "streamed_GBps": (rows_b + cols_b + 32.0 * n + (4 + 2 * m) * 8.0 * P) / (ms * 1e-3) / 1e9,
               "streamed_frac_of_hbm_peak": (rows_b + cols_b + 32.0 * n + (4 + 2 * m) * 8.0 * P) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "shard": f"{n} samples x {k} {dist} columns of {D} features, logistic, m=10",
               "what": "gdmix_fe_eval + gdmix_fe_step per L-BFGS evaluation; bytes = 16 B/nnz (value + index, both passes) + 32 B/sample "
                       "+ (4 + 2m) x 8 B/coefficient"}
    del packed
    torch.cuda.empty_cache()
    return out


def cli_end_to_end_leg(entities):
    """SURVEY.md 8(f) N3 + rows a9-a11 next to the headline: the drop-in CLI on files. C2-shaped entities written as entity-grouped
    TFRecord partitions of 125 k (the reference's input format), `gdmix_amd.gdmix --stage=random_effect --action=train` in this
    process (device context and libraries already up: the steady state of a long job), model + score Avro out; then again as a
    warm start from the model files just written. tools/e2e_bench.py is the same run with the time of every phase."""
    import logging
    import shutil
    import tempfile
    from gdmix_amd import gdmix as cli
    from gdmix_amd import synthetic
    from gdmix_amd.io.grouped_reader import write_grouped_partition
    parts = max(1, entities // 125_000)
    md = {"features": [{"name": "bag", "dtype": "float", "shape": [1024], "isSparse": True},
                       {"name": "offset", "dtype": "float", "shape": [], "isSparse": False},
                       {"name": "uid", "dtype": "long", "shape": [], "isSparse": False},
                       {"name": "ent", "dtype": "string", "shape": [], "isSparse": False}],
          "labels": [{"name": "response", "dtype": "int", "shape": [], "isSparse": False}]}
    logging.getLogger("gdmix_amd").setLevel(logging.WARNING)
    with tempfile.TemporaryDirectory() as d:
        b = synthetic.make_batch(entities, 16, 4, 1024, seed=1)
        per = (entities + parts - 1) // parts
        for k in range(parts):
            sub = b.select(np.arange(k * per, min(entities, (k + 1) * per)))
            write_grouped_partition(os.path.join(d, "train", "active", f"partitionId={k}", "part-0.tfrecord"), sub, "ent", "bag",
                                    weight_column_name=None)
        with open(os.path.join(d, "meta.json"), "w") as f:
            json.dump(md, f)
        with open(os.path.join(d, "features.csv"), "w") as f:
            f.write("".join(f"f{i},\n" for i in range(1024)))
        with open(os.path.join(d, "plist.txt"), "w") as f:
            f.write(",".join(str(k) for k in range(parts)))
        size_in = sum(os.path.getsize(os.path.join(r, x)) for r, _, fs in os.walk(os.path.join(d, "train")) for x in fs)
        argv = ["gdmix", "--stage=random_effect", "--model_type=logistic_regression", "--uid_column_name=uid",
                "--label_column_name=response", "--prediction_score_column_name=predictionScore",
                f"--partition_list_file={d}/plist.txt", f"--training_data_dir={d}/train", f"--metadata_file={d}/meta.json",
                f"--output_model_dir={d}/models", "--feature_bag=bag", f"--feature_file={d}/features.csv",
                "--partition_entity=ent", "--regularize_bias=False", "--l2_reg_weight=1.0", f"--training_score_dir={d}/ts",
                "--action=train"]
        os.environ.pop("TF_CONFIG", None)
        cold_runs, warm_runs = [], []
        for rep in range(1 + HOST_LEG_REPEATS):   # a first cold run to settle, then (cold, warm start from its models) pairs
            for kind in ("cold", "warm"):
                if kind == "cold" and os.path.isdir(os.path.join(d, "models")):
                    shutil.rmtree(os.path.join(d, "models"))
                    shutil.rmtree(os.path.join(d, "ts"), ignore_errors=True)
                if rep == 0 and kind == "warm":
                    continue
                t = time.perf_counter()
                cli.run(argv)
                if rep:
                    (cold_runs if kind == "cold" else warm_runs).append(time.perf_counter() - t)
        times = [None, float(np.median(cold_runs)), float(np.median(warm_runs))]
        size_out = sum(os.path.getsize(os.path.join(r, x)) for r, _, fs in os.walk(d) for x in fs if x.endswith(".avro"))
        # the same run the way gdmix-workflow starts a stage (single_node/local_ops.py:42-54): a fresh `python -m ...` child per
        # stage — interpreter start, imports, HIP context, library load and the run itself; files are in the page cache
        import subprocess
        sub = {}
        child = {"cold": [], "warm_start": []}
        for rep in range(HOST_LEG_REPEATS):
            for label, wipe in (("cold", True), ("warm_start", False)):
                if wipe:
                    shutil.rmtree(os.path.join(d, "models"))
                    shutil.rmtree(os.path.join(d, "ts"), ignore_errors=True)
                t = time.perf_counter()
                # (the child logs at INFO as the reference does; its stderr is kept out of this run's and shown only if it fails)
                cp = subprocess.run([sys.executable, "-m", "gdmix_amd.gdmix"] + argv[1:], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                    env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")))
                child[label].append(time.perf_counter() - t)
                if cp.returncode != 0:
                    raise RuntimeError(f"python -m gdmix_amd.gdmix exited with {cp.returncode}: " + cp.stderr.decode(errors="replace")[-2000:])
        for label, xs in child.items():
            sub[label + "_s"] = float(np.median(xs))
            sub[label + "_s_spread"] = _spread(xs)
            sub[label + "_entities_per_s"] = entities / sub[label + "_s"]
        t = time.perf_counter()
        subprocess.check_call([sys.executable, "-c", "import gdmix_amd.gdmix"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        sub["import_only_s"] = time.perf_counter() - t
        sub["what"] = ("wall time of a real child process `python -m gdmix_amd.gdmix --stage=random_effect --action=train ...` on the same "
                       "files (cold = no prior model; warm_start = from the cold run's model files); import_only_s = a child that only imports the CLI module")
    return {"entities": entities, "partitions": parts, "tfrecord_bytes_in": size_in, "avro_bytes_out": size_out,
            "cold_s": times[1], "cold_entities_per_s": entities / times[1], "warm_start_s": times[2],
            "warm_start_entities_per_s": entities / times[2],
            "cold_entities_per_s_spread": _spread(entities / t for t in cold_runs), "warm_start_entities_per_s_spread": _spread(entities / t for t in warm_runs),
            "what": "python -m gdmix_amd.gdmix --stage=random_effect --action=train on entity-grouped TFRecord partitions: decode, "
                    "upload, pack, solve, score the training data, model + score Avro files; in-process second run (cold = no prior "
                    "model), then a warm start from its model files"}, sub


def cli_shape_leg(kind, entities):
    """The drop-in CLI on partition directories of BASELINE's other shapes: kind = "c5" — Zipf-sized entities
    (SURVEY 8(d) sizes, up to 2^20 non-zeros, D = 65 536) — or "ml20m_movie" — the MovieLens-20M per-movie population (head of
    54 k samples). Entities are hashed into 8 partitions (Java hash, as DataPartitioner does), written as entity-grouped TFRecords,
    trained in this process (second pass timed: context and libraries up), model + score Avro out, phases timed by wrapping the
    model's own methods (summed thread time of each: reads ahead and writes behind overlap the device work)."""
    import logging
    import shutil
    import tempfile
    from gdmix_amd import gdmix as cli
    from gdmix_amd import model as model_mod
    from gdmix_amd import synthetic
    from gdmix_amd.partition_dirs import write_partition_dir
    logging.getLogger("gdmix_amd").setLevel(logging.WARNING)
    t0 = time.perf_counter()
    if kind == "c5":
        b = synthetic.make_survey_batch(entities, 32, 8, 65536, seed=synthetic.C5_SEED, size_dist="c5zipf", with_uid=True)
        dim = 65536
    else:
        import bench_strong
        b = bench_strong.ml20m_population("per_movie", None)
        if b.uid is None:
            b.uid = np.arange(b.N, dtype=np.int64)
        dim = 24
    t_gen = time.perf_counter() - t0
    phases = {}
    M = model_mod.RandomEffectLRLBFGSModel
    saved = []

    def timed(owner, name, label):
        fn = getattr(owner, name)
        saved.append((owner, name, fn))

        def wrapper(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                phases[label] = phases.get(label, 0.0) + time.perf_counter() - t
        setattr(owner, name, wrapper)
    with tempfile.TemporaryDirectory() as d:
        t = time.perf_counter()
        argv, members, size_in = write_partition_dir(d, b, 8, dim)
        t_write = time.perf_counter() - t
        os.environ.pop("TF_CONFIG", None)
        try:
            timed(M, "_read", "read_tfrecord_s")
            timed(M, "_solve_batch", "upload_pack_solve_readback_s")
            timed(M, "_save_model", "model_avro_s")
            timed(model_mod, "_write_scores", "score_avro_s")
            times, runs, phase_runs = [], [], []
            for rep in range(1 + HOST_LEG_REPEATS):   # the first run settles (context, libraries, page cache); the median of the others is the figure
                phases.clear()
                if rep:
                    shutil.rmtree(os.path.join(d, "models"))
                    shutil.rmtree(os.path.join(d, "ts"), ignore_errors=True)
                t = time.perf_counter()
                cli.run(argv)
                (runs if rep else times).append(time.perf_counter() - t)
                phase_runs.append(dict(phases))
            mid = int(np.argsort(runs)[len(runs) // 2])
            times.append(runs[mid])
            phases = phase_runs[1 + mid]
        finally:
            for owner, name, fn in saved:
                setattr(owner, name, fn)
        size_out = sum(os.path.getsize(os.path.join(r, x)) for r, _, fs in os.walk(d) for x in fs if x.endswith(".avro"))
    z = b.ent_nnz()
    dom = max(phases, key=phases.get) if phases else None
    return {"shape": kind, "entities": int(b.E), "samples": int(b.N), "nnz": int(b.Z), "largest_entity_nnz": int(z.max()), "partitions": len(members),
            "tfrecord_bytes_in": size_in, "avro_bytes_out": size_out, "cold_s": times[1], "entities_per_s": b.E / times[1],
            "entities_per_s_spread": _spread(b.E / t for t in runs),
            "first_run_s": times[0], "phases_thread_s": {k: round(v, 4) for k, v in phases.items()}, "dominant_phase": dom,
            "generate_s": round(t_gen, 2), "write_tfrecord_s": round(t_write, 2),
            "what": "python -m gdmix_amd.gdmix --stage=random_effect --action=train (in process, second run) on 8 Java-hashed partitions of this "
                    "shape: TFRecord decode, upload, pack, solve, scoring of the training data, model + score Avro"}


def lbfgs_state_bytes(p, nit, nfev, opts_m=10, n=None, one_workgroup=False):
    """What the compact-form L-BFGS itself moves per entity when its state does not fit on chip (re_solve_team.hpp), an ESTIMATE
    from the iteration counts: every evaluation reads the stored pairs once for the 2m products fused into the gradient's
    epilogue, every iteration reads them once more for the direction and writes the new pair (16 B per pair and coefficient), and
    the five p-vectors x, g, d, t, r are read and written once per evaluation (16 B per vector and coefficient) — unless they sit
    in the LDS a one-workgroup entity's workgroup leaves free (csrc/re_solve.hip: team_vec_level — five vectors and the residuals
    if they fit into 15 290 doubles, else three, else x alone). pairs at iteration k = min(k, m). Calibration on Zipf-1M
    (profiles/r05_zipf_1m.txt): 8 + 213 GB estimated for the one-workgroup class against 82.6 GB FETCH_SIZE (x 2 for its 16-byte
    loads, MI355X_MICROARCH.md) + 25.1 GB WRITE_SIZE = 190 GB counted."""
    p, nit, nfev = np.asarray(p, np.float64), np.asarray(nit, np.float64), np.asarray(nfev, np.float64)
    full = np.maximum(nit - opts_m, 0.0)
    pairs_sum = np.minimum(nit, opts_m) * (np.minimum(nit, opts_m) + 1.0) / 2.0 + full * opts_m      # sum over iterations of the pairs stored
    mean_pairs = np.where(nit > 0, pairs_sum / np.maximum(nit, 1.0), 0.0)
    vectors = np.full(p.shape, 5.0)
    if one_workgroup and n is not None:
        n = np.asarray(n, np.float64)
        arena = 15290.0
        vectors = np.where(5 * p + n <= arena, 0.0, np.where(3 * p + n <= arena, 2.0, np.where(p + n <= arena, 4.0, 5.0)))
    return float((p * (16.0 * pairs_sum + 16.0 * mean_pairs * nfev + 16.0 * nit + 16.0 * vectors * nfev)).sum())


def chain_leg():
    """SURVEY.md 8(f) N2 next to the headline: one pass of the coordinate chain through the drop-in CLI (gdmix_amd/chain.py) —
    global fixed effect -> per-user -> per-movie random effect on MovieLens-100K-shaped data with planted effects, the partition
    job's offset update (previous stage's float scores joined on uid) and DataPartitioner's active / passive bounding in between.
    In this process, second pass (context and libraries up); tests/test_gpu_chain.py holds the same chain to the CPU oracle's."""
    import tempfile
    from gdmix_amd import chain
    data = chain.make_dataset()
    outs = []
    for _ in range(1 + HOST_LEG_REPEATS):
        with tempfile.TemporaryDirectory() as d:
            outs.append(chain.run_chain(d, data, num_partitions=4, upper_bounds={"per_user": 48}))
    outs = sorted(outs[1:], key=lambda o: o["total_s"])
    out = outs[len(outs) // 2]
    return {"total_s_spread": _spread(o["total_s"] for o in outs), "fe_s": out["global"]["s"], "per_user_s": out["per_user"]["s"], "per_movie_s": out["per_movie"]["s"],
            "partition_s": out["per_user"]["partition_s"] + out["per_movie"]["partition_s"], "total_s": out["total_s"],
            "validation_auc": [out[s]["validation_auc"] for s in chain.STAGES], "train_auc": [out[s]["train_auc"] for s in chain.STAGES],
            "samples": out["global"]["train_samples"] + out["global"]["validation_samples"],
            "what": "python -m gdmix_amd.gdmix x 3 (fixed_effect, random_effect per user with an active-data bound of 48, random_effect per movie) + the "
                    "partition step between stages, 100 k ratings of 943 users x 1 682 movies, in process, second pass; AUC per stage on the validation split"}


class Workload:
    """What a step runs on: a raw batch in HBM (`raw_dev`, the dict REDeviceSolver.pack takes) plus the per-entity host arrays
    the accounting needs (samples, non-zeros, label sums) and a way to get some entities as a host RawBatch (CPU leg)."""

    def __init__(self, name, raw_dev, n, z, ones, host=None, take=None, what=""):
        self.name, self.raw_dev, self.n, self.z, self.ones, self.host, self._take, self.what = name, raw_dev, n, z, ones, host, take, what
        self.E, self.N, self.Z = int(n.size), int(n.sum()), int(z.sum())

    def host_sample(self, count):
        ents = np.arange(min(int(count), self.E))
        return self.host.select(ents) if self.host is not None else self._take(ents)


def make_workload(a, rank, solver):
    from gdmix_amd import synthetic
    w = a.workload
    if w == "c5share":
        import torch
        raw, n = synthetic.make_c5_share_device(solver.device, a.c5_entities, seed=synthetic.C5_SEED + rank)
        ptr = raw["ent_row_ptr"]
        cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=ptr.device), torch.cumsum(raw["y"].double(), 0)])
        ones = (cs[ptr[1:]] - cs[ptr[:-1]]).cpu().numpy()
        return Workload(w, raw, n, n * (raw["Z"] // raw["N"]), ones, take=lambda ents: synthetic.device_entities_to_host(raw, n, ents),
                        what=f"C5 per-GPU share: {a.c5_entities} entities/GPU, nnz Zipf-like P(nnz>=x)~x^-1.2 on [8, 2^20] rescaled to mean 256, "
                             "k=8 distinct uniform columns of D=65536 (SURVEY 8(d) generator, on the device), per-entity L2 LR, L-BFGS m=10")
    if w == "c2":
        batch = synthetic.make_survey_batch(a.entities, a.mean_n, a.k, a.dim, seed=synthetic.C2_SEED + rank, entity_id_base=rank * a.entities)
        what = (f"C2: synthetic {a.entities} entities/GPU x avg {a.mean_n * a.k} nnz (n~max(1,Poisson({a.mean_n})), k={a.k} distinct uniform "
                f"columns of D={a.dim}, SURVEY 8(d) generator), per-entity L2 LR, L-BFGS m=10")
    elif w == "c5mean":
        batch = synthetic.make_batch(a.entities, 32, 8, 65536, seed=synthetic.C5_SEED + rank, with_uid=False)
        what = f"exploration shape {w}, {a.entities} entities/GPU"
    elif w == "zipf":
        batch = synthetic.make_batch(a.entities, 32, 8, 65536, seed=synthetic.C5_SEED + rank, size_dist="zipf", with_uid=False)
        what = f"exploration shape {w}, {a.entities} entities/GPU"
    elif w in ("ml_user", "ml_movie"):
        batch = synthetic.make_movielens_like(a.entities, "per_user" if w == "ml_user" else "per_movie", seed=100 + rank)
        what = f"exploration shape {w}, {a.entities} entities/GPU"
    else:
        kind = "per_user" if w == "ml20m_user" else "per_movie"
        if rank == 0:     # the same population the strong-scaling / projection legs split (generated once per process)
            import bench_strong
            batch = bench_strong.ml20m_population(kind, None)
        else:
            batch = synthetic.make_movielens_20m(kind, seed=200 + rank)
        what = (f"C3: MovieLens-20M-sized {kind} random effect, {batch.E} entities/GPU, {batch.N} training rows (count distributions of the public "
                "dataset card, feature bags of scripts/download_process_movieLens_data.py; synthetic values), per-entity L2 LR, L-BFGS m=10")
    ones = np.add.reduceat(batch.y.astype(np.float64), batch.ent_row_ptr[:-1]) if batch.N else np.zeros(batch.E)
    return Workload(w, solver.upload(batch), batch.ent_n(), batch.ent_nnz(), ones, host=batch, what=what)


def other_workloads_leg(a, rank, world, solver, opts, coll_dev):
    """BASELINE.json's other random-effect configurations at their own entity sizes, next to the C2 headline: C3 = MovieLens-20M
    per-user and per-movie (north_star: "entities-converged/sec on synthetic MovieLens-shaped data"), C5 = one GPU's share of the
    Zipf-sized job. One warm-up and two timed steps (pack + solve) each, every rank on its own data (weak scaling), max over
    ranks; the same code path as `--workload X`, which also prints the roofline of X's dominant kernel."""
    import copy
    import torch
    import torch.distributed as dist
    out = {}
    for w in ("ml20m_user", "ml20m_movie", "c5share"):
        b = copy.copy(a)
        b.workload = w
        if w == "c5share":
            # 4 M Zipf entities take ~125 GB of a device (raw arrays, two alternating pack workspaces, results): skipped — by all
            # ranks together — when a rank does not have that much free, e.g. ranks sharing one device in the harness test
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            free = torch.cuda.mem_get_info()[0]
            sharing = world if getattr(a, "ranks_share_device", False) else 1   # ranks that will ask the same device for it
            ok = 1 if free >= int(150e9 * b.c5_entities / 4_000_000) * sharing else 0
            if world > 1:
                tt = torch.tensor([ok], dtype=torch.int64, device=coll_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MIN)
                ok = int(tt.item())
            if not ok:
                if rank == 0:
                    out[w] = {"skipped": "not enough free device memory on some rank (%.0f GB free on rank 0, %d rank(s) per device)" % (free / 1e9, sharing)}
                continue
        t_gen = time.perf_counter()
        wl = make_workload(b, rank, solver)
        t_gen = time.perf_counter() - t_gen
        packed = res = None
        for _ in range(2):   # warm-up: kernels, and the allocator's cached blocks
            packed = res = None      # the previous step's workspace (35 GB for C5) goes back to the allocator before the next is asked for
            packed = solver.pack(wl.raw_dev)
            res = solver.solve(packed, opts)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        steps = 2
        t0 = time.perf_counter()
        for _ in range(steps):
            packed = res = None
            packed = solver.pack(wl.raw_dev)
            res = solver.solve(packed, opts)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        kernel_ms_timed = np.array(solver.last_solve_ms())
        # launch durations for the roofline: one more step with the classes one after another (in the timed steps the large classes
        # run side by side and stretch each other: gdmix_re_set_spread)
        solver.set_spread(0)
        try:
            packed = res = None
            packed = solver.pack(wl.raw_dev)
            res = solver.solve(packed, opts)
            torch.cuda.synchronize()
            kernel_ms = np.array(solver.last_solve_ms())
        finally:
            solver.set_spread(SPREAD_DEFAULT)
        st = res.status
        conv = int(((st >= 0) & (st <= 2)).sum().item())
        if world > 1:
            tt = torch.tensor([dt, float(conv)], dtype=torch.float64, device=coll_dev)
            mx = tt.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(tt)
            dt, conv = float(mx[0].item()), int(tt[1].item())
        if rank == 0:
            classes = solver.class_counts(packed)
            top = sorted(((float(kernel_ms[c]), classes[c][0], int(classes[c][1])) for c in range(len(classes)) if kernel_ms[c] > 0), reverse=True)[:4]
            n = wl.n
            # the HBM roofline of this workload's dominant launch, by the same formulas as the headline's (B(e), and the
            # re-streamed figure, which is the algorithmic one for entities that do not stay on chip: the team kernels)
            p_e = np.diff(packed.coef_ptr_host())
            cls_e = packed._view(packed.c.cls_tmp, packed.E, torch.int32).cpu().numpy()
            dom_c = int(np.argmax(kernel_ms))
            if classes[dom_c][0].startswith("re_solve_tall_kernel<1> p<=64") and classes[dom_c - 1][0].endswith("lean p<=64") and kernel_ms[dom_c - 1] == 0:
                cls_e = np.where(cls_e == dom_c - 1, dom_c, cls_e)      # a small lean class ran inside the general launch
            sel = cls_e == dom_c
            nfev_e = res.nfev.cpu().numpy().astype(np.float64)
            b_alg = float((8.0 * wl.z[sel] + 16.0 * n[sel] + 8.0 * p_e[sel] + 32.0).sum())
            b_str = float((nfev_e[sel] * (8.0 * wl.z[sel] + 16.0 * n[sel]) + 8.0 * p_e[sel] + 32.0).sum())
            b_state = lbfgs_state_bytes(p_e[sel], res.nit.cpu().numpy()[sel], nfev_e[sel], opts_m=10, n=n[sel], one_workgroup=classes[dom_c][0].endswith("workgroup"))
            dms = float(kernel_ms[dom_c])
            roof = {"kernel": classes[dom_c][0], "entities_in_launch": int(sel.sum()), "avg_launch_ms": dms,
                    "alg_bytes_per_launch": b_alg, "achieved_GBps": b_alg / (dms * 1e-3) / 1e9, "frac_of_hbm_peak": b_alg / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "restreamed_bytes_per_launch": b_str, "restreamed_GBps": b_str / (dms * 1e-3) / 1e9,
                    "restreamed_frac_of_hbm_peak": b_str / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, "mean_nfev": float(nfev_e[sel].mean()),
                    "lbfgs_state_bytes_per_launch": b_state, "restreamed_plus_state_GBps": (b_str + b_state) / (dms * 1e-3) / 1e9,
                    "restreamed_plus_state_frac_of_hbm_peak": (b_str + b_state) / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "state_note": "for entities that do not stay on chip (team kernels) the L-BFGS state is streamed as well: per evaluation the m-pair history "
                                  "once for the products fused into the gradient and per iteration once more for the direction (16 B x pairs x p each) plus the five "
                                  "p-vectors — lbfgs_state_bytes() below; for the resident classes (group kernels) it never leaves the CU and the figure does not apply",
                    "avg_launch_ms_in_the_timed_steps": float(kernel_ms_timed[dom_c]),
                    "note": "B(e) of SURVEY 8(d) over the launch's entities / its HIP-event duration in one more step with the size classes one after another "
                            "(gdmix_re_set_spread 0; side by side, as in the timed steps, launches stretch each other); restreamed = nfev x (8 nnz + 16 n) + 8 p + 32"}
            out[w] = {"what": wl.what, "entities_per_gpu": wl.E, "N": wl.N, "Z": wl.Z, "entities_per_s": conv * steps / dt,
                      "ms_per_step": dt / steps * 1e3, "converged_per_step": conv, "host_generate_s": t_gen,
                      "parity_classes": {"W": int(((wl.ones > 0) & (wl.ones < n)).sum()), "D": int(wl.E - ((wl.ones > 0) & (wl.ones < n)).sum())},
                      "mean_nit": res.nit.double().mean().item(), "mean_nfev": res.nfev.double().mean().item(),
                      "largest_launches": [{"kernel": k, "entities": e, "ms": round(ms, 3)} for ms, k, e in top], "roofline": roof}
        del wl, packed, res
        torch.cuda.empty_cache()
    return out if rank == 0 else None


def _fits(w, a, world, coll_dev):
    """The c5 population needs ~150 GB per 4 M entities of a rank's share (raw arrays, pack workspace, results, and for the re-balanced
    run the wire form): skipped — by all ranks together — when a rank does not have that, e.g. ranks sharing one device in the tests."""
    if w != "c5":
        return True
    import torch
    import torch.distributed as dist
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info()[0]
    sharing = world if getattr(a, "ranks_share_device", False) else 1
    per_rank = 150e9 if (world == 1 or a.no_rebalance) else 200e9    # the give-back of a re-balanced run assembles the results once more
    ok = 1 if free >= int(per_rank * a.c5_entities / 4_000_000) * sharing else 0
    if world > 1:
        tt = torch.tensor([ok], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        ok = int(tt.item())
    return bool(ok)


def strong_main(a, rank, world, solver, opts, coll_dev, backend):
    """--scaling strong: the headline is ONE population split over the ranks (bench_strong.py)."""
    import bench_strong
    name = a.workload if a.workload in bench_strong.STRONG_WORKLOADS else "c5"
    r = bench_strong.strong_leg(name, rank, world, solver, opts, coll_dev, a.c5_entities, steps=a.steps, warmup=a.warmup,
                                rebalance=not a.no_rebalance, ml_entities=a.ml_entities or None, tolerance=a.rebalance_tolerance)
    if rank == 0:
        line = {"metric": "random-effect entities converged/sec", "value": round(r["entities_per_s"], 1), "unit": "entities/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": r["ms"], "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": r["what"], "workload_key": name, "total_entities": r["total_entities"], "step": "pack+solve",
                           "l2": 1.0, "regularize_bias": False, "max_iter": 100, "parallelism": f"partitions[rank::{world}]",
                           "collective_backend": backend},
                "roofline": None, "cpu_baseline": None,
                "note": "strong-scaling mode: roofline and cpu_baseline belong to the default (weak, C2) line",
                "strong_scaling": [r]}
        emit(line, a)


DETAIL_DEFAULT = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
RESULT_FD = None        # the process's original stdout once main() has pointed fd 1 at stderr
COMPACT_LIMIT = 4096   # the driver keeps the tail of stdout: the result line has to be short (BENCH_r04: a 38.5 KB line was cut, parsed = null)


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or d.get(k) is None:
            return None
        d = d[k]
    return d


def _r(x, digits=4):
    """Numbers of the short line: `digits` significant digits."""
    if isinstance(x, bool) or x is None or isinstance(x, (str, int)):
        return x
    try:
        return float(f"{float(x):.{digits}g}")
    except (TypeError, ValueError):
        return x


def compact_line(full, detail_file):
    """The ONE stdout line of a run: the bench contract's keys, `roofline` and `cpu_baseline` as flat objects, one number per
    side leg under `summary`, and the path of the file that holds everything else (`full`, what earlier rounds printed as one
    line). Always shorter than COMPACT_LIMIT bytes: optional parts are dropped, last first, until it is."""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    line["ms_per_step"] = _r(line["ms_per_step"], 6)
    cfg = full.get("config") or {}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:330]}
    if _get(cfg, "library", "build_id"):
        line["config"]["build_id"] = cfg["library"]["build_id"]
    for k in ("workload_key", "entities_per_gpu", "total_entities", "step", "parallelism", "collective_backend"):
        if cfg.get(k) is not None:
            line["config"][k] = cfg[k]
    if cfg.get("ranks"):
        line["config"]["rank_ms_per_step"] = [_r(r["ms_per_step"]) for r in cfg["ranks"]]
    roof = full.get("roofline")
    if roof:
        line["roofline"] = {"bound": roof["bound"], "achieved": _r(roof["achieved"], 6), "peak": roof["peak"], "unit": roof["unit"],
                            "frac": _r(roof["frac"]), "traffic": roof.get("traffic"), "kernel": roof.get("kernel"),
                            "avg_launch_ms": _r(roof.get("avg_launch_ms")), "alg_bytes_per_launch": roof.get("alg_bytes_per_launch"),
                            "entities_in_launch": roof.get("entities_in_launch"),
                            "alone_frac": _r(_get(roof, "alone", "frac")), "alone_launch_ms": _r(_get(roof, "alone", "avg_launch_ms")),
                            "valu_issue_frac": _r(_get(roof, "valu", "issue_frac")),
                            "traffic_over_alg": _r(roof["traffic"] / roof["alg_bytes_per_launch"]) if roof.get("traffic") and roof.get("alg_bytes_per_launch") else None}
        if roof.get("traffic") is None and roof.get("traffic_note"):
            line["roofline"]["traffic_note"] = str(roof["traffic_note"])[:160]
    else:
        line["roofline"] = None
    cpu = full.get("cpu_baseline")
    line["cpu_baseline"] = ({"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"], "sample": str(cpu["sample"])[:220],
                             "reference_quoted": cpu.get("reference_quoted")} if cpu else None)
    d = full.get("detail") or {}
    s = {}

    def put(key, value, digits=4):
        if value is not None:
            s[key] = _r(value, digits)
    put("pack_ms", d.get("pack_ms_per_step"))
    put("solve_ms", d.get("solve_ms_per_step"))
    put("converged_per_step", d.get("converged_per_step"), 12)
    put("mean_nfev", d.get("mean_nfev"))
    put("handover_eps", _get(d, "host_handover", "entities_per_s"))
    put("score_frac", _get(d, "score_pass", "frac_of_hbm_peak"))
    put("fe_frac", _get(d, "fixed_effect_eval", "frac_of_hbm_peak"))
    put("fe_ms_per_eval", _get(d, "fixed_effect_eval", "ms_per_evaluation"))
    put("fe_streamed_frac", _get(d, "fixed_effect_eval", "streamed_frac_of_hbm_peak"))
    put("fe_zipf_frac", _get(d, "fixed_effect_eval", "zipf", "frac_of_hbm_peak"))
    put("fe_zipf_ms_per_eval", _get(d, "fixed_effect_eval", "zipf", "ms_per_evaluation"))
    put("solve_to_host_ms", _get(d, "solve_to_host", "ms_per_step"))
    for w in ("ml20m_user", "ml20m_movie", "c5share"):
        put(w + "_ms", _get(d, "workloads", w, "ms_per_step"))
        put(w + "_eps", _get(d, "workloads", w, "entities_per_s"))
    put("c5share_dom_restreamed_frac", _get(d, "workloads", "c5share", "roofline", "restreamed_frac_of_hbm_peak"))
    put("c5share_dom_stream_plus_state_frac", _get(d, "workloads", "c5share", "roofline", "restreamed_plus_state_frac_of_hbm_peak"))
    put("cli_cold_eps", _get(d, "cli_end_to_end", "cold_entities_per_s"))
    put("cli_warm_eps", _get(d, "cli_end_to_end", "warm_start_entities_per_s"))
    put("cli_child_s", _get(d, "cli_subprocess", "cold_s"))
    put("cli_c5_eps", _get(d, "cli_end_to_end_c5", "entities_per_s"))
    put("cli_ml20m_movie_eps", _get(d, "cli_end_to_end_ml20m_movie", "entities_per_s"))
    for p in d.get("strong_projection") or []:
        s.setdefault("proj8", {})[p["workload"] + "_ms"] = _r(p.get("ms"))
        if p.get("ms_mean") is not None:
            s["proj8"][p["workload"] + "_ms_mean"] = _r(p["ms_mean"])
    full_share = d.get("c5_full_share")
    if full_share and "skipped" not in full_share:
        s["c5_full_share"] = {k: _r(full_share.get(k)) for k in ("entities", "s", "entities_per_s", "serial_s", "round_ms_p50", "round_ms_p99", "round_ms_max",
                                                                 "imbalance_vs_8_rank_mean_round") if full_share.get(k) is not None}
        pb = full_share.get("partitions_per_batch") or {}
        if "8" in pb:
            s["c5_full_share"]["eps_8_partitions_per_batch"] = _r(pb["8"]["entities_per_s"])
        pr = full_share.get("projected_rounds")
        if pr:
            s["c5_full_share"].update(rounds=len(pr["rounds"]), plain_ms_total=_r(pr["plain_ms_total"]), rebalanced_ms_total=_r(pr["rebalanced_ms_total"]),
                                      imbalance=_r(pr["mean_imbalance"]), imbalance_after=_r(pr["mean_imbalance_after"]))
    chain = d.get("chain")
    if chain:
        s["chain"] = {k: _r(v) for k, v in chain.items() if isinstance(v, (int, float))}
        s["chain"]["validation_auc"] = [_r(x) for x in chain.get("validation_auc", [])]
    # the host-bound legs are medians of HOST_LEG_REPEATS runs: [min, max] of each next to them
    spread = {}
    for key, path in (("handover_eps", ("host_handover", "entities_per_s_spread")), ("cli_cold_eps", ("cli_end_to_end", "cold_entities_per_s_spread")),
                      ("cli_warm_eps", ("cli_end_to_end", "warm_start_entities_per_s_spread")), ("cli_child_s", ("cli_subprocess", "cold_s_spread")),
                      ("cli_c5_eps", ("cli_end_to_end_c5", "entities_per_s_spread")), ("cli_ml20m_movie_eps", ("cli_end_to_end_ml20m_movie", "entities_per_s_spread")),
                      ("chain_total_s", ("chain", "total_s_spread"))):
        sp = _get(d, *path)
        if sp:
            spread[key] = [_r(sp["min"], 3), _r(sp["max"], 3)]
    if spread:
        s["host_legs_min_max"] = spread
    line["summary"] = s
    st = full.get("strong_scaling")
    if st:
        line["strong"] = {x["workload"]: {"ms": _r(x.get("ms")), "entities_per_s": _r(x.get("entities_per_s")), "imbalance": _r(x.get("imbalance")),
                                          "rebalanced_ms": _r(_get(x, "rebalanced", "ms"))} for x in st}
    if full.get("note"):
        line["note"] = str(full["note"])[:160]
    line["detail_file"] = detail_file
    # never longer than the limit: drop the optional parts, least important first
    for drop in (("note",), ("cpu_baseline", "sample"), ("summary", "host_legs_min_max"), ("summary", "proj8"), ("strong",), ("summary",), ("config", "rank_ms_per_step")):
        if len(json.dumps(line)) < COMPACT_LIMIT:
            break
        tgt = line
        for k in drop[:-1]:
            tgt = tgt.get(k) or {}
        tgt.pop(drop[-1], None)
    line["config"]["workload"] = line["config"]["workload"][:max(40, 330 - max(0, len(json.dumps(line)) - COMPACT_LIMIT + 1))]
    return line


def gpu_state():
    """Clocks, power and temperature of device 0 as rocm-smi reports them at the end of the run (a per-movie step of
    6.18 ms on the driver's box against 5.30 on the builder's had no explanation on record — the next such gap can be read against
    the state of the box). Best effort: {} when the tool is missing or its output changes."""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        out = subprocess.run([exe, "-d", "0", "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"], capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(out.stdout).values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(w in kl for w in ("sclk", "mclk", "fclk", "power", "temperature (sensor junction)", "performance level")):
                keep[k] = v
        return keep
    except Exception as e:  # noqa: BLE001
        return {"unavailable": str(e)[:120]}


def emit(full, a):
    """Write everything to the detail file and print the short line LAST on stdout (the only stdout line of the run)."""
    path = a.detail_file or DETAIL_DEFAULT
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(full, fh)
    except OSError as e:      # a read-only checkout: the short line still goes out
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
        path = None
    if a.print_detail:
        print(json.dumps(full), file=sys.stderr)
    sys.stderr.flush()
    if os.environ.get("GDMIX_BENCH_LINE") == "full":      # tools/ scripts that pipe the full result (never the driver's command)
        text = json.dumps(full)
    else:
        shown = path and (os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else os.path.abspath(path))
        text = json.dumps(compact_line(full, shown))
    sys.stdout.flush()
    if RESULT_FD is None:
        print(text, flush=True)
    else:
        os.write(RESULT_FD, (text + "\n").encode())


def main():
    a = parse()
    import logging
    logging.disable(logging.INFO)      # the model logs every partition at INFO, as the reference does: not into the bench's output
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)      # does not return
    # stdout carries the result line and nothing else: whatever a library prints there (gloo's "[Gloo] Rank 0 is connected ...",
    # a child process) goes to stderr from here on; emit() writes to the saved descriptor
    global RESULT_FD
    if RESULT_FD is None:
        sys.stdout.flush()
        RESULT_FD = os.dup(1)
        os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} was launched with WORLD_SIZE={world}: start one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...)")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback path)")
    if a.ranks_share_device:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: HIP device {local_rank} is not visible ({torch.cuda.device_count()} devices)")
    torch.cuda.set_device(local_rank)
    coll_dev = "cpu" if a.ranks_share_device else "cuda"    # where the few scalars of the collectives live
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if a.ranks_share_device else "nccl"
        if a.ranks_share_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from gdmix_amd import build
    if rank == 0:
        build.build_library()
    if world > 1:
        dist.barrier()
    from gdmix_amd.solver import NUM_CLASSES, REDeviceSolver, SolverOptions

    opts_kw = dict(l2=1.0, regularize_bias=False, has_intercept=True, m=a.lbfgs_m, max_iter=100, ftol=1e-12)
    opts = SolverOptions(**opts_kw)
    solver = REDeviceSolver(local_rank)
    if a.scaling == "strong":
        strong_main(a, rank, world, solver, opts, coll_dev, backend)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if a.workload == "c5":
        raise SystemExit("--workload c5 is a strong-scaling population: use --scaling strong (c5share is its weak-scaling counterpart)")
    t_gen = time.perf_counter()
    wl = make_workload(a, rank, solver)
    t_gen = time.perf_counter() - t_gen
    batch = wl.host
    raw_dev = wl.raw_dev
    packed = solver.pack(raw_dev)
    out = solver.alloc_result(packed)
    solver.set_timing(True)
    if a.giant_nnz >= 0:
        solver.set_giant_nnz(a.giant_nnz)
    if a.team_nnz >= 0:
        solver.set_team_nnz(a.team_nnz)
    if a.tall_min_n >= 0:
        solver.set_tall_min_n(a.tall_min_n)
    if a.kernel_mask >= 0:
        solver.set_kernel_mask(a.kernel_mask)
    if a.tall_split_n >= 1:
        solver.set_tall_split_n(a.tall_split_n)

    kernel_ms = np.zeros(NUM_CLASSES)
    ev_pack = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    pack_ms = solve_ms = 0.0
    step_wall = []

    def measured_step():
        """One step as it is timed: pack + solve, bracketed by events, the per-class kernel times read back. The warm-up runs the
        same body (its numbers are dropped): the first use of the timing events and of gdmix_re_last_solve_ms costs ~4 ms of host
        time once per process (BENCH r04: step 1 of 3 took 14.8 ms of wall for 10.8 ms of device work), which is not a step's."""
        nonlocal packed, pack_ms, solve_ms, kernel_ms
        ev_pack[0].record()
        if not a.solve_only:
            packed = solver.pack(raw_dev)
        ev_pack[1].record()
        r = solver.solve(packed, opts, out=out)
        ev_pack[2].record()
        kernel_ms += np.array(solver.last_solve_ms())      # waits for this step's solve kernels
        ev_pack[2].synchronize()                            # (recorded right behind them; not necessarily complete yet)
        pack_ms += ev_pack[0].elapsed_time(ev_pack[1])
        solve_ms += ev_pack[1].elapsed_time(ev_pack[2])
        step_wall.append((time.perf_counter(), ev_pack[0].elapsed_time(ev_pack[1]), ev_pack[1].elapsed_time(ev_pack[2])))
        return r

    for _ in range(a.warmup):
        measured_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    kernel_ms = np.zeros(NUM_CLASSES)
    pack_ms = solve_ms = 0.0
    step_wall = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = measured_step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0          # this rank's K steps
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0              # ... and until the slowest rank is done
    status = res.status
    converged = int(((status >= 0) & (status <= 2)).sum().item())
    per_rank = None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        mine = torch.tensor([rank, local_rank, dt_own / a.steps * 1e3, converged, wl.E], dtype=torch.float64, device=coll_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": int(t[0]), "device": int(t[1]), "ms_per_step": float(t[2]), "converged_per_step": int(t[3]), "entities": int(t[4])}
                    for t in (x.cpu() for x in allr)]
        converged_all = sum(r["converged_per_step"] for r in per_rank)
    else:
        converged_all = converged
    value = converged_all * a.steps / dt
    # clocks and power WHILE the device works (outside the timed region): rocm-smi is asked from a second thread while this one keeps
    # solving the same batch; the state at the end of the whole run (idle clocks) is recorded too
    gpu_load_state = None
    if rank == 0 and world == 1:
        import threading
        box = {}
        th = threading.Thread(target=lambda: box.update(state=gpu_state()))
        th.start()
        n_extra = 0
        while th.is_alive() and n_extra < 400:
            solver.solve(packed, opts, out=out)
            n_extra += 1
        th.join()
        torch.cuda.synchronize()
        gpu_load_state = dict(box.get("state") or {}, solves_meanwhile=n_extra)
    # ---- the same step with the size classes one after another (gdmix_re_set_spread 0), NOT part of `value`: in the timed region the
    # large classes run side by side on four queues (their tails overlap: the step is shorter) and every launch lasts longer than it
    # would alone — a kernel's own roofline figure needs its duration alone
    saved = (kernel_ms.copy(), pack_ms, solve_ms, list(step_wall))
    alone_cls_ms, alone_solve_ms = kernel_ms / max(1, a.steps), solve_ms / max(1, a.steps)    # (--no-alone, or spread off: the timed steps themselves)
    if not a.no_alone and SPREAD_DEFAULT > 1:
        kernel_ms = np.zeros(NUM_CLASSES)
        solve_ms = 0.0
        solver.set_spread(0)
        try:
            measured_step()
            kernel_ms = np.zeros(NUM_CLASSES)
            solve_ms = 0.0
            for _ in range(3):
                res = measured_step()
            torch.cuda.synchronize()
        finally:
            solver.set_spread(SPREAD_DEFAULT)
        alone_cls_ms, alone_solve_ms = kernel_ms / 3.0, solve_ms / 3.0
    kernel_ms, pack_ms, solve_ms, step_wall = saved[0], saved[1], saved[2], saved[3]
    # ---- the single-GPU legs next to the headline, BEFORE the large workloads: measured after them (BENCH_r03) the hand-over ran at
    # 58 M entities/s instead of 77 M — the 125 GB of the C5 share churn the allocator's blocks and the page-locked staging
    # (profiles/r04_host_path.txt: the same leg alone, with and without round 3's side stream)
    p_host = int(packed.P)
    th_host = out["theta_thr"]
    c2_legs = a.workload == "c2" and world == 1
    # SURVEY.md §8(d) metric (i) to the letter: "resident ragged-CSR batch -> theta resident on host". The headline step ends with
    # theta in HBM (config.step says so); this is the same step plus the copy of the thresholded coefficients (float64) and the
    # status words into page-locked host memory, serial on one stream (the hand-over leg below overlaps it with the next partition)
    to_host = None
    if c2_legs and not a.no_e2e:
        th_pin = torch.empty(p_host, dtype=torch.float64).pin_memory()
        st_pin = torch.empty(wl.E, dtype=torch.int32).pin_memory()
        evh = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ms = []
        for i in range(1 + 3):
            evh[0].record()
            pk = packed if a.solve_only else solver.pack(raw_dev)
            r = solver.solve(pk, opts, out=out)
            th_pin.copy_(r.theta_thr[:p_host], non_blocking=True)
            st_pin.copy_(r.status, non_blocking=True)
            evh[1].record()
            evh[1].synchronize()
            if i:
                ms.append(evh[0].elapsed_time(evh[1]))
        to_host = {"ms_per_step": float(np.median(ms)), "runs_ms": [round(x, 3) for x in ms], "d2h_bytes": p_host * 8 + wl.E * 4,
                   "entities_per_s": wl.E / (float(np.median(ms)) * 1e-3),
                   "what": "pack + solve + D2H of theta_thr (f64) and status into page-locked memory, one stream, no overlap"}
        del th_pin, st_pin
    # host hand-over (SURVEY.md §8(d) metric (ii)): packed host batch -> H2D -> solve -> D2H -> thresholded theta on the host
    e2e = None
    if c2_legs and not a.no_e2e:
        e2e = host_handover(batch, opts, local_rank, p_host)
    # the scoring pass over the same resident batch (the path's HBM-bound stream; not part of `value`)
    score = None
    if not a.no_e2e and world == 1:
        th = th_host
        for _ in range(2):
            solver.score(packed, th)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10):
            solver.score(packed, th)
        ev[1].record()
        torch.cuda.synchronize()
        sms = ev[0].elapsed_time(ev[1]) / 10
        sbytes = 8.0 * wl.Z + 16.0 * wl.N + 4.0 * (wl.N + wl.E) + 8.0 * float(packed.P) + 24.0 * wl.E
        score = {"ms": sms, "samples_per_s": wl.N / (sms * 1e-3), "alg_bytes": sbytes, "GBps": sbytes / (sms * 1e-3) / 1e9,
                 "frac_of_hbm_peak": sbytes / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "what": "gdmix_re_score: logits of every sample, 8 B/nnz + 16 B/sample + 8 B/coefficient + pointers"}
    fe_eval = None
    if c2_legs and not a.no_fe:
        fe_eval = fixed_effect_leg(solver, a.fe_rows)
        fe_eval["zipf"] = fixed_effect_leg(solver, a.fe_rows, dist="zipf")   # the driver sees the skewed shard too
    cli_e2e = cli_sub = cli_c5 = cli_movie = chain_res = None
    if c2_legs and not a.no_cli:
        cli_e2e, cli_sub = cli_end_to_end_leg(a.cli_entities)
        chain_res = chain_leg()
        if not a.no_other_workloads:
            cli_c5 = cli_shape_leg("c5", a.cli_c5_entities)
            cli_movie = cli_shape_leg("ml20m_movie", None)
    others = None
    if a.workload == "c2" and not a.no_other_workloads:
        del res
        res = None
        others = other_workloads_leg(a, rank, world, solver, opts, coll_dev)
        res = solver.solve(packed, opts, out=out)    # (the statistics of the C2 batch again, for the lines below)

    # ---- ONE population split over the ranks (BASELINE configs[2] and [4]): measured with ranks, projected on one GPU ----
    strong = projection = None
    if a.workload == "c2" and not a.no_other_workloads:
        import bench_strong
        res = None
        torch.cuda.empty_cache()
        if world > 1 and not a.no_strong:
            strong = [bench_strong.strong_leg(w, rank, world, solver, opts, coll_dev, a.c5_entities, steps=a.strong_steps, warmup=2,
                                              rebalance=not a.no_rebalance, ml_entities=a.ml_entities or None, tolerance=a.rebalance_tolerance)
                      for w in bench_strong.STRONG_WORKLOADS if _fits(w, a, world, coll_dev)]
        elif world == 1 and a.project_ranks > 1:
            projection = [bench_strong.projection_leg(w, a.project_ranks, solver, opts, a.c5_entities, steps=a.strong_steps, warmup=2,
                                                      ml_entities=a.ml_entities or None, tolerance=a.rebalance_tolerance)
                          for w in bench_strong.STRONG_WORKLOADS if _fits(w, a, world, coll_dev)]
        solver.set_timing(True)
        res = solver.solve(packed, opts, out=out)
    c5_full = None
    if a.workload == "c2" and world == 1 and not a.no_other_workloads and a.c5_full_entities > 0:
        import bench_strong
        res = None
        torch.cuda.empty_cache()
        need = 70e9 * a.c5_full_entities / 100_000_000 + 8e9
        if torch.cuda.mem_get_info()[0] >= need:
            c5_full = bench_strong.c5_full_share_leg(solver, opts, local_rank, a.c5_full_entities, ranks=8, partitions=1024,
                                                     projected_rounds=a.c5_full_rounds, tolerance=a.rebalance_tolerance)
        else:
            c5_full = {"skipped": "not enough free device memory (%.0f GB free, %.0f GB needed)" % (torch.cuda.mem_get_info()[0] / 1e9, need / 1e9)}
        solver.set_timing(True)
        res = solver.solve(packed, opts, out=out)

    if rank == 0:
        # ---- roofline of the dominant kernel = the size-class launch with the largest share of a step ----
        n, z = wl.n, wl.z
        p = np.diff(packed.coef_ptr_host())
        b_e = 8.0 * z + 16.0 * n + 8.0 * p + 32.0                    # B(e), SURVEY.md §8(d)
        alg_bytes = float(b_e.sum())
        # parity classes of SURVEY.md §8(d): W = both labels present (a finite optimum exists with the unregularised intercept),
        # D = all labels equal (the solver runs until the gradient test passes; only invariants are comparable)
        well_posed = int(((wl.ones > 0) & (wl.ones < n)).sum())
        classes = solver.class_counts(packed)
        cls = packed._view(packed.c.cls_tmp, packed.E, torch.int32).cpu().numpy()
        cls_ms = kernel_ms / a.steps
        # a class that has entities and no launch of its own ran with the class behind it (a small lean tall class, re_api.hip)
        pending = []
        for c in range(len(classes)):
            if cls_ms[c] > 0:
                for q in pending:
                    cls[cls == q] = c
                pending = []
            elif classes[c][1] > 0:
                pending.append(c)
        dom = int(np.argmax(cls_ms))
        dom_bytes = float(b_e[cls == dom].sum())
        dom_ms = float(cls_ms[dom])
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        dom_alone_ms = float(alone_cls_ms[dom])
        achieved_alone = dom_bytes / (dom_alone_ms * 1e-3) / 1e9 if dom_alone_ms > 0 else 0.0
        all_ms = solve_ms / a.steps      # the solve call's device time (the classes overlap: their launch durations do not add up)
        nfev = res.nfev.double().mean().item()
        nit = res.nit.double().mean().item()
        nfev_e = res.nfev.cpu().numpy().astype(np.float64)
        nit_e = res.nit.cpu().numpy().astype(np.float64)
        # the re-streamed figure of SURVEY.md §8(d), B_stream(e) = nfev (8 nnz + 16 n) + 8 p + 32: what a design that does not keep
        # the entity resident moves — and for the classes whose entities do not fit on chip (team kernels) the algorithmic figure
        b_s = nfev_e * (8.0 * z + 16.0 * n) + 8.0 * p + 32.0
        b_stream = float(b_s.sum())
        per_class = []
        for c in np.flatnonzero(cls_ms > 0):
            sel = cls == c
            per_class.append({"kernel": classes[c][0], "entities": int(sel.sum()), "ms": round(float(cls_ms[c]), 3),
                              "alg_GBps": round(float(b_e[sel].sum()) / (cls_ms[c] * 1e-3) / 1e9, 2),
                              "restreamed_GBps": round(float(b_s[sel].sum()) / (cls_ms[c] * 1e-3) / 1e9, 2),
                              "lbfgs_state_GB": round(lbfgs_state_bytes(p[sel], nit_e[sel], nfev_e[sel], a.lbfgs_m, n=n[sel], one_workgroup=classes[c][0].endswith("workgroup")) / 1e9, 3)
                              if "team" in classes[c][0] else None,
                              "mean_n": round(float(n[sel].mean()), 1), "mean_p": round(float(p[sel].mean()), 1),
                              "max_nnz": int(z[sel].max()), "mean_nfev": round(float(nfev_e[sel].mean()), 2)})
        traffic = traffic_detail = valu = None
        traffic_note = None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        workload_id = f"{a.workload} E={wl.E} n~{a.mean_n} k={a.k} D={a.dim} survey-generator" if a.workload == "c2" else f"{a.workload} E={wl.E}"
        # PMC counters of the same kernel from separate rocprofv3 --pmc runs of this configuration (tools/profile_round.sh);
        # they are per launch of a size class and only valid for the workload they were collected on
        if not os.path.exists(tpath):
            traffic_note = "profiles/latest_traffic.json is missing: run tools/profile_round.sh on the GPU box"
        else:
            with open(tpath) as fh:
                tj = json.load(fh)
            from gdmix_amd import build as _b
            lib_abi = int(solver.lib.gdmix_re_abi_version())
            if tj.get("_re_abi") != lib_abi or tj.get("_re_kernel_sources_sha16") != _b.kernel_source_hash():
                traffic_note = (f"profiles/latest_traffic.json is stale: collected on ABI {tj.get('_re_abi')} / kernel sources {tj.get('_re_kernel_sources_sha16')}, "
                                f"this library is ABI {lib_abi} / {_b.kernel_source_hash()} (tools/profile_round.sh collects it again)")
            elif tj.get("_workload") != workload_id:
                traffic_note = f"PMC passes were collected on workload {tj.get('_workload')!r}, this run is {workload_id!r}"
            elif classes[dom][0] not in tj:
                traffic_note = f"no PMC pass holds the dominant class {classes[dom][0]!r}"
            else:
                traffic_detail = tj[classes[dom][0]]
                traffic = traffic_detail["bytes"]
                if traffic_detail.get("valu_insts"):
                    # the binding resource: VALU issue. Peak = CUs x 4 SIMDs x clock / 4 cycles per wave64
                    # fp64-path instruction.
                    insts = float(traffic_detail["valu_insts"])
                    peak = 256 * 4 * 2.4e9 / 4.0
                    valu = {"insts_per_launch": insts, "insts_per_entity": insts / max(1, int(classes[dom][1])),
                            "issue_rate_Ginst_s": insts / (dom_alone_ms * 1e-3) / 1e9 if dom_alone_ms else None,
                            "issue_frac": insts / (dom_alone_ms * 1e-3) / peak if dom_alone_ms else None,
                            "peak_Ginst_s": peak / 1e9,
                            "source": "SQ_INSTS_VALU (rocprofv3 --pmc, separate pass) / live launch duration of the kernel alone (roofline.alone); peak = 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles"}
        roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note, "traffic_detail": traffic_detail,
                    "valu": valu,
                    "kernel": classes[dom][0], "entities_in_launch": int(classes[dom][1]),
                    "avg_launch_ms": dom_ms, "alg_bytes_per_launch": dom_bytes,
                    "schedule": f"timed region: the large size classes run side by side on {SPREAD_DEFAULT} queues (gdmix_re_set_spread), so this launch shares the "
                                "device with the other classes' and lasts longer than alone; `alone` = the same launch with the classes one after another",
                    "alone": {"avg_launch_ms": dom_alone_ms, "achieved": round(achieved_alone, 3), "frac": achieved_alone / HBM_PEAK_GBS,
                              "solve_ms_per_step": alone_solve_ms, "class_ms": [round(float(x), 3) for x in alone_cls_ms],
                              "what": "the timed steps themselves (--no-alone, or GDMIX_RE_SPREAD <= 1)" if (a.no_alone or SPREAD_DEFAULT <= 1)
                                      else "3 untimed steps after the timed region with gdmix_re_set_spread(ctx, 0)"},
                    "alg_bytes_per_entity": alg_bytes / wl.E,
                    "restreamed_bytes_per_launch": float(b_s[cls == dom].sum()),
                    "all_solve_kernels": {"ms_per_step": all_ms, "alg_GBps": alg_bytes / (all_ms * 1e-3) / 1e9 if all_ms else 0.0},
                    "note": "LDS/register-resident L-BFGS: bound by fp64 VALU issue and latency, not by HBM "
                            "(SURVEY.md §8d honesty note); see DESIGN.md for the VALU-side accounting"}
        cpu = None
        if not a.no_cpu_baseline and world == 1:   # the CPU leg is timed at N = 1 only
            # a bounded sample: about 13 M non-zeros per pass (200 k entities of C2)
            per_ent = max(1.0, wl.Z / wl.E)
            sample = a.cpu_sample if a.cpu_sample > 0 else int(min(wl.E, max(64, 12.8e6 / per_ent)))
            sub = wl.host_sample(sample)
            v, cores, es, d, passes = cpu_baseline(sub, opts_kw)
            cpu = {"value": round(v, 1), "unit": "entities/s", "cores": cores, "kind": "port",
                   # the reference itself (TF + scipy, one entity per fmin_l_bfgs_b call) cannot run on the GPU box; its rate measured in the build
                   # container on these shapes is quoted, not timed here
                   "reference_quoted": {"value": 639, "unit": "entities/s/core", "source": "BASELINE.md section 3"},
                   "sample": f"first {es} entities of the same {a.workload} batch ({sub.Z} non-zeros) x {passes} passes, oracle/re_oracle.c fp64 "
                             f"(restatement pinned to the reference by tests/golden), {cores} threads, {d:.1f} s"}
        line = {
            "metric": "random-effect entities converged/sec", "value": round(value, 1), "unit": "entities/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl.what, "workload_key": a.workload,
                       "entities_per_gpu": wl.E, "step": "solve, theta on device" if a.solve_only else "pack+solve, theta on device",
                       "l2": 1.0, "regularize_bias": False, "max_iter": 100, "parallelism": f"entity-shard x{world}",
                       "collective_backend": backend, "ranks": per_rank,
                       # which binary ran: the hash of its sources (gdmix_re_build_id) and of its compiler flags
                       "library": {"build_id": solver.lib.gdmix_re_build_id().decode(), "flags_id": build.embedded_id(build.LIB, build.FLAGS_MARKER),
                                   "sources_id": build.source_id()}},
            "roofline": roofline, "cpu_baseline": cpu, "strong_scaling": strong,
            "detail": {"strong_projection": projection, "step_wall_ms": [round((b[0] - c) * 1e3, 3) for b, c in zip(step_wall, [t0] + [x[0] for x in step_wall[:-1]])],
                       "step_device_ms": [[round(x[1], 3), round(x[2], 3)] for x in step_wall],
                       "pack_ms_per_step": pack_ms / a.steps, "solve_ms_per_step": solve_ms / a.steps,
                       "unique_compaction": ("next to the solve (gdmix_re_set_defer_unique): inside solve_ms, not pack_ms"
                                             if os.environ.get("GDMIX_RE_DEFER_UNIQUE", "1") != "0" else "inside the pack"),
                       "solve_kernel_ms_per_step": float(kernel_ms.sum()) / a.steps,
                       "classes": classes, "class_ms": [round(float(x) / a.steps, 3) for x in kernel_ms], "per_class": per_class,
                       "mean_nit": nit, "mean_nfev": nfev,
                       "converged_per_step": converged_all, "parity_classes": {"W": well_posed, "D": int(wl.E - well_posed)}, "N": wl.N, "Z": wl.Z, "P": packed.P,
                       "gpu_state_under_load": gpu_load_state, "gpu_state_at_end": gpu_state(),
                       "host_generate_s": t_gen, "solve_to_host": to_host, "host_handover": e2e, "score_pass": score, "fixed_effect_eval": fe_eval, "cli_end_to_end": cli_e2e,
                       "cli_subprocess": cli_sub, "cli_end_to_end_c5": cli_c5, "cli_end_to_end_ml20m_movie": cli_movie, "workloads": others, "c5_full_share": c5_full, "chain": chain_res,
                       "restreamed_bytes_per_step": b_stream,
                       "restreamed_GBps": b_stream / (float(kernel_ms.sum()) / a.steps * 1e-3) / 1e9 if kernel_ms.sum() else None},
        }
        emit(line, a)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
