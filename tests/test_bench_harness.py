"""bench.py's launch contract: `python bench.py --gpus N` starts N ranks itself (one process per GPU under
torch.distributed.run), and refuses loudly when the box has fewer devices — it must never print n_gpus = 1 for a run that
was asked for N."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--entities", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-fe", "--no-cli", "--no-other-workloads"]


def _run(args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TF_CONFIG"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True,
                          timeout=timeout)


def _line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_gpus_flag_without_enough_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the devices")
    r = _run(["--gpus", "2"] + SMALL, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "device" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_world_size_must_match_gpus_flag():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL, env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_gpus_flag_starts_that_many_ranks():
    """--gpus 2 without a launcher: two ranks. On the 1-GPU box they share cuda:0 over gloo (harness test, numbers meaningless);
    with two devices visible they are two RCCL ranks on two GPUs."""
    import torch
    share = [] if torch.cuda.device_count() >= 2 else ["--ranks-share-device"]
    line = _line(_run(["--gpus", "2"] + share + SMALL))
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    ranks = line["config"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1]
    assert [r["device"] for r in ranks] == ([0, 0] if share else [0, 1])
    assert line["config"]["collective_backend"] == ("gloo" if share else "nccl")
    assert all(r["entities"] == 20000 and r["converged_per_step"] == 20000 and r["ms_per_step"] > 0 for r in ranks)
    assert line["detail"]["converged_per_step"] == 40000
    assert abs(line["value"] - 40000 * line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) < 1e-6 * line["value"]
    assert line["ms_per_step"] >= max(r["ms_per_step"] for r in ranks) * 0.999


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["ml20m_movie"])
def test_other_workloads_produce_a_line(workload):
    line = _line(_run(["--workload", workload, "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-e2e"]))
    assert line["n_gpus"] == 1 and line["config"]["workload_key"] == workload
    assert line["detail"]["converged_per_step"] == line["config"]["entities_per_gpu"] == 26744
    pc = line["detail"]["per_class"]
    assert sum(c["entities"] for c in pc) == 26744 and len(pc) >= 4


@pytest.mark.gpu
def test_default_legs_of_a_two_rank_run():
    """The driver's scaling run is the default command with --gpus N: every rank also measures the MovieLens-20M workloads and
    C5's per-GPU share after C2 (detail.workloads, collectives inside). Two ranks here; on the 1-GPU box they share the device and
    the C5 leg must be skipped by both ranks together (a rank that ran out of memory alone would leave the other in a barrier)."""
    import torch
    share = [] if torch.cuda.device_count() >= 2 else ["--ranks-share-device"]
    line = _line(_run(["--gpus", "2"] + share + ["--entities", "20000", "--steps", "2", "--warmup", "1"], timeout=1500))
    assert line["n_gpus"] == 2 and line["detail"]["converged_per_step"] == 40000
    w = line["detail"]["workloads"]
    assert set(w) == {"ml20m_user", "ml20m_movie", "c5share"}
    assert w["ml20m_user"]["entities_per_gpu"] == 138493 and w["ml20m_user"]["converged_per_step"] == 2 * 138493
    assert w["ml20m_movie"]["entities_per_gpu"] == 26744 and w["ml20m_movie"]["entities_per_s"] > 0
    assert ("skipped" in w["c5share"]) == bool(share)
    if not share:
        assert w["c5share"]["converged_per_step"] == 2 * w["c5share"]["entities_per_gpu"]
    # ... and ONE population of each kind split over the two ranks (top-level strong_scaling; the C5 population is skipped together
    # with c5share when the ranks share a device)
    st = {s["workload"]: s for s in line["strong_scaling"]}
    assert set(st) == ({"ml20m_user", "ml20m_movie"} if share else {"ml20m_user", "ml20m_movie", "c5"})
    assert st["ml20m_user"]["total_entities"] == 138493 == sum(r["entities"] for r in st["ml20m_user"]["per_rank"])
    assert st["ml20m_movie"]["total_entities"] == 26744 and st["ml20m_movie"]["rebalanced"]["status_equal"] == 26744
    assert st["ml20m_user"]["rebalanced"]["max_rel_diff_vs_plain"] <= 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("workload,extra,total", [("ml20m_user", ["--ml-entities", "6000"], 6000), ("ml20m_movie", ["--ml-entities", "3000"], 3000),
                                                  ("c5", ["--c5-entities", "30000"], 60000)])
def test_strong_scaling_splits_one_population_and_rebalancing_returns_the_same_models(workload, extra, total):
    """--scaling strong: ONE population, entity -> partition by the Java hash, partition -> rank by partitions[rank::2]; the same
    share then through the re-balancer (exchange -> widen -> pack -> solve -> give back), whose coefficients must be the plain
    run's. Two ranks: on the 1-GPU box they share cuda:0 and the collectives are staged over gloo; with two devices it is RCCL."""
    import torch
    share = [] if torch.cuda.device_count() >= 2 else ["--ranks-share-device"]
    line = _line(_run(["--gpus", "2"] + share + ["--scaling", "strong", "--workload", workload, "--steps", "2", "--warmup", "1"] + extra))
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["config"]["workload_key"] == workload
    s = line["strong_scaling"][0]
    assert s["total_entities"] == total == sum(r["entities"] for r in s["per_rank"]) == s["converged_per_step"]
    assert all(r["entities"] > 0 and r["ms_per_step"] > 0 for r in s["per_rank"]) and s["imbalance"] >= 1.0
    assert sum(r["partitions"] for r in s["per_rank"]) <= s["partitions"]
    assert abs(line["value"] - total / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    rb = s["rebalanced"]
    assert rb["status_equal"] == total and rb["max_rel_diff_vs_plain"] <= 1e-7
    assert sum(r["bytes_sent"] for r in rb["per_rank"]) == sum(r["bytes_received"] for r in rb["per_rank"])
    assert all(r["solve_ms"] > 0 for r in rb["per_rank"])
    if workload == "c5":      # ... and at the product path's granularity: one partition per worker and round, priced by the rounds before
        pr = s["partition_rounds"]
        assert len(pr["rounds"]) >= 2 and pr["rounds"][0]["priced_by"] == "non-zeros" and pr["rounds"][1]["priced_by"].startswith("measured")
        assert all(r["max_rel_diff_vs_plain"] <= 1e-7 and len(r["plain_ms"]) == 2 for r in pr["rounds"])


@pytest.mark.gpu
def test_projection_of_an_eight_rank_job_on_one_device():
    """--gpus 1: the shares of an 8-rank job solved one after another (detail.strong_projection), with the re-balancing plan."""
    line = _line(_run(["--entities", "20000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-fe", "--no-cli",
                       "--ml-entities", "8000", "--c5-entities", "20000", "--project-ranks", "8"], timeout=1500))
    proj = {p["workload"]: p for p in line["detail"]["strong_projection"]}
    assert set(proj) == {"ml20m_user", "ml20m_movie", "c5"} and line["strong_scaling"] is None
    assert proj["ml20m_user"]["total_entities"] == 8000 and proj["c5"]["total_entities"] == 160000
    for p in proj.values():
        assert p["ranks"] == 8 and len(p["per_rank"]) == 8 and p["imbalance"] >= 1.0
        assert sum(r["converged"] for r in p["per_rank"]) == p["total_entities"]
        assert abs(p["ms"] - max(r["ms_per_step"] for r in p["per_rank"])) < 1e-9
        plan = p["rebalance_plan"]
        assert plan["after_imbalance"] <= plan["predicted_imbalance"] + 1e-9
