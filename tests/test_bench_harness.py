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
