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


def _run(args, timeout=900, detail=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TF_CONFIG"):
        env.pop(k, None)
    if detail is not None:
        args = list(args) + ["--detail-file", detail]
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True,
                          timeout=timeout)


def _line(args, tmp_path, timeout=900):
    """Run bench.py; check the output contract — stdout is ONE short JSON line (the driver keeps only the tail of stdout) that names
    the file holding the full result — and return the full result with the short line under "_short"."""
    detail = str(tmp_path / "bench_detail.json")
    r = _run(args, timeout=timeout, detail=detail)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = r.stdout.splitlines()
    assert len(out) == 1 and out[0].startswith("{"), r.stdout[-2000:]
    assert len(out[0]) < 4096
    short = json.loads(out[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "detail_file"):
        assert k in short, k
    assert "INFO:" not in r.stderr      # the model's per-partition log lines stay out of the bench's output
    with open(detail) as fh:
        full = json.load(fh)
    assert full["value"] == short["value"] and full["n_gpus"] == short["n_gpus"]
    full["_short"] = short
    return full


def test_short_line_of_a_full_default_run_fits_the_driver():
    """The full result of round 4's default run was a 38.5 KB line and the driver recorded `parsed: null`. The short form of that
    very result: under 4 KB, the contract's keys, roofline and cpu_baseline flat, one number per side leg."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r04_final_bench_line.json")) as fh:
        full = json.load(fh)
    assert len(json.dumps(full)) > 30000
    short = bench.compact_line(full, "gpurun_out/bench_detail.json")
    text = json.dumps(short)
    assert len(text) < bench.COMPACT_LIMIT == 4096 and json.loads(text) == short
    assert short["value"] == full["value"] and short["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert short["config"]["workload"].startswith("C2") and "model" not in short["config"]
    rf = short["roofline"]
    assert rf["bound"] == "hbm" and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3) and rf["traffic"] > rf["alg_bytes_per_launch"]
    assert rf["kernel"].startswith("re_solve_grp_kernel") and rf["alone_frac"] > rf["frac"] and 0 < rf["valu_issue_frac"] < 1
    assert short["cpu_baseline"]["kind"] == "port" and short["cpu_baseline"]["cores"] >= 1
    sm = short["summary"]
    assert sm["fe_frac"] == pytest.approx(full["detail"]["fixed_effect_eval"]["frac_of_hbm_peak"], rel=1e-3)
    assert set(sm["proj8"]) >= {"ml20m_user_ms", "ml20m_movie_ms", "c5_ms"} and sm["c5share_ms"] > 0 and sm["cli_cold_eps"] > 0
    # round 6's keys, grafted onto that result: where theta ends, the step with the copy to the host, the reference's
    # quoted CPU rate, the Zipf fixed-effect shard and the [min, max] of every repeated host-bound leg
    sp = {"median": 3.5e6, "min": 3.1e6, "max": 3.9e6, "runs": [3.1e6, 3.5e6, 3.9e6]}
    full["detail"]["solve_to_host"] = {"ms_per_step": 19.25}
    full["detail"]["fixed_effect_eval"]["zipf"] = {"frac_of_hbm_peak": 0.47, "ms_per_evaluation": 0.58}
    full["detail"]["fixed_effect_eval"]["streamed_frac_of_hbm_peak"] = 0.455
    full["detail"]["cli_end_to_end"]["cold_entities_per_s_spread"] = sp
    full["detail"]["host_handover"]["entities_per_s_spread"] = sp
    full["cpu_baseline"]["reference_quoted"] = {"value": 639, "unit": "entities/s/core", "source": "BASELINE.md section 3"}
    full["config"]["step"] = "pack+solve, theta on device"
    short = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert len(json.dumps(short)) < 4096
    assert short["summary"]["solve_to_host_ms"] == 19.25 and short["summary"]["fe_zipf_frac"] == 0.47 and short["summary"]["fe_streamed_frac"] == 0.455
    assert short["summary"]["host_legs_min_max"] == {"handover_eps": [3.1e6, 3.9e6], "cli_cold_eps": [3.1e6, 3.9e6]}
    assert short["cpu_baseline"]["reference_quoted"]["value"] == 639 and short["config"]["step"].endswith("theta on device")
    # ... and whatever a run adds, the line stays short: optional parts go first
    full["config"]["workload"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    full["config"]["ranks"] = [{"ms_per_step": 1.2345678 + i} for i in range(512)]
    fat = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert len(json.dumps(fat)) < 4096 and fat["value"] == full["value"] and fat["roofline"]["frac"] == rf["frac"]


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
def test_gpus_flag_starts_that_many_ranks(tmp_path):
    """--gpus 2 without a launcher: two ranks. On the 1-GPU box they share cuda:0 over gloo (harness test, numbers meaningless);
    with two devices visible they are two RCCL ranks on two GPUs."""
    import torch
    share = [] if torch.cuda.device_count() >= 2 else ["--ranks-share-device"]
    line = _line(["--gpus", "2"] + share + SMALL, tmp_path)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    ranks = line["config"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1]
    assert [r["device"] for r in ranks] == ([0, 0] if share else [0, 1])
    assert line["config"]["collective_backend"] == ("gloo" if share else "nccl")
    # the short line the driver parses carries both (N > 1 stays one command away)
    assert len(line["_short"]["config"]["rank_ms_per_step"]) == 2 and line["_short"]["config"]["collective_backend"] == ("gloo" if share else "nccl")
    assert all(r["entities"] == 20000 and r["converged_per_step"] == 20000 and r["ms_per_step"] > 0 for r in ranks)
    assert line["detail"]["converged_per_step"] == 40000
    assert abs(line["value"] - 40000 * line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) < 1e-6 * line["value"]
    assert line["ms_per_step"] >= max(r["ms_per_step"] for r in ranks) * 0.999


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["ml20m_movie"])
def test_other_workloads_produce_a_line(workload, tmp_path):
    line = _line(["--workload", workload, "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-e2e"], tmp_path)
    assert line["n_gpus"] == 1 and line["config"]["workload_key"] == workload
    assert line["detail"]["converged_per_step"] == line["config"]["entities_per_gpu"] == 26744
    pc = line["detail"]["per_class"]
    assert sum(c["entities"] for c in pc) == 26744 and len(pc) >= 4


@pytest.mark.gpu
def test_default_legs_of_a_two_rank_run(tmp_path):
    """The driver's scaling run is the default command with --gpus N: every rank also measures the MovieLens-20M workloads and
    C5's per-GPU share after C2 (detail.workloads, collectives inside). Two ranks here; on the 1-GPU box they share the device and
    the C5 leg must be skipped by both ranks together (a rank that ran out of memory alone would leave the other in a barrier)."""
    import torch
    share = [] if torch.cuda.device_count() >= 2 else ["--ranks-share-device"]
    line = _line(["--gpus", "2"] + share + ["--entities", "20000", "--steps", "2", "--warmup", "1"], tmp_path, timeout=1500)
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
def test_strong_scaling_splits_one_population_and_rebalancing_returns_the_same_models(workload, extra, total, tmp_path):
    """--scaling strong: ONE population, entity -> partition by the Java hash, partition -> rank by partitions[rank::2]; the same
    share then through the re-balancer (exchange -> widen -> pack -> solve -> give back), whose coefficients must be the plain
    run's. Two ranks: on the 1-GPU box they share cuda:0 and the collectives are staged over gloo; with two devices it is RCCL."""
    import torch
    share = [] if torch.cuda.device_count() >= 2 else ["--ranks-share-device"]
    line = _line(["--gpus", "2"] + share + ["--scaling", "strong", "--workload", workload, "--steps", "2", "--warmup", "1"] + extra, tmp_path)
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
def test_projection_of_an_eight_rank_job_on_one_device(tmp_path):
    """--gpus 1: the shares of an 8-rank job solved one after another (detail.strong_projection), with the re-balancing plan."""
    line = _line(["--entities", "20000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-fe", "--no-cli",
                  "--ml-entities", "8000", "--c5-entities", "20000", "--project-ranks", "8", "--c5-full-entities", "400000", "--c5-full-rounds", "3"],
                 tmp_path, timeout=1500)
    proj = {p["workload"]: p for p in line["detail"]["strong_projection"]}
    assert set(proj) == {"ml20m_user", "ml20m_movie", "c5"} and line["strong_scaling"] is None
    assert proj["ml20m_user"]["total_entities"] == 8000 and proj["c5"]["total_entities"] == 160000
    for p in proj.values():
        assert p["ranks"] == 8 and len(p["per_rank"]) == 8 and p["imbalance"] >= 1.0
        assert sum(r["converged"] for r in p["per_rank"]) == p["total_entities"]
        assert abs(p["ms"] - max(r["ms_per_step"] for r in p["per_rank"])) < 1e-9
        plan = p["rebalance_plan"]
        assert plan["after_imbalance"] <= plan["predicted_imbalance"] + 1e-9
    # C5 at the product path's granularity: worker 0's 128 partitions of ONE population (here 400 k entities), one partition per round,
    # serial and over three contexts; the first rounds of the whole 8-worker job plain and with the re-balancing plan applied
    fs = line["detail"]["c5_full_share"]
    assert fs["partitions"] == 128 and fs["converged"] == fs["converged_pipelined"] == fs["entities"] and 40_000 < fs["entities"] < 60_000
    assert len(fs["round_ms"]) == 128 and fs["round_ms_p50"] <= fs["round_ms_p99"] <= fs["round_ms_max"] and fs["s"] > 0 and fs["serial_s"] > 0
    assert all(fs["partitions_per_batch"][k]["converged"] == fs["entities"] for k in ("2", "4", "8"))
    pr = fs["projected_rounds"]
    assert len(pr["rounds"]) == 3 and pr["rounds"][0]["priced_by"] == "non-zeros" and pr["rounds"][1]["priced_by"].startswith("measured")
    assert all(len(r["plain_ms"]) == 8 and len(r["rebalanced_ms"]) == 8 and r["imbalance"] >= 1.0 for r in pr["rounds"])
    short = line["_short"]["summary"]["c5_full_share"]
    assert short["entities"] == fs["entities"] and short["rounds"] == 3
