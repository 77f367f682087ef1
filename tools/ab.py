#!/usr/bin/env python3
"""One parametrised A/B runner for bench.py (replaces round 4's thirty one-off r04_*.sh scripts).

    python tools/ab.py [--env NAME=v1,v2,...]... [--lib a.so,b.so] [--workloads c2,c5share] [--reps 2]
                       [--fields ms_per_step,detail.pack_ms_per_step,...] [--tests "-k expr"] [--out DIR] [-- bench args]

Every combination of the --env values (and of --lib, copied over gdmix_amd/libgdmix_re.so) x every workload x `reps` rounds runs
bench.py with the side legs off; the chosen fields of the full result (the detail file) are printed per run, then the per-variant
minimum. `--tests` first runs the GPU parity tests matching the expression. `unset` as a value leaves the variable out.

Recipes (what round 4's scripts were; DESIGN.md / docs/rounds cite the results):
    spread of the size classes     --env GDMIX_RE_SPREAD=0,2,3,4 --workloads c2,c5share
    hardware queues per process    --env GPU_MAX_HW_QUEUES=4,8,16 --legs e2e --fields ms_per_step,detail.host_handover.entities_per_s
    same-XCD team barrier          --env GDMIX_RE_XCD_BARRIER=0,1 --workloads zipf,c5share --tests "team_tiers or device_wide"
    LDS vectors of the team kernel --env GDMIX_TEAM_ARENA_KB=0,200 --workloads zipf,c5share
    bitmap pack                    --env GDMIX_PACK_BITMAP=0,1 --workloads c2,ml20m_user --tests "pack"
    tall team class                --env GDMIX_RE_TALL_TEAM=0,1 --workloads ml20m_user,ml20m_movie
    history pairs (direction cost) --workloads c2 -- --lbfgs-m 1     (and again with --lbfgs-m 10)
    two builds                     --lib gdmix_amd/lib_a.so,gdmix_amd/lib_b.so
"""
import argparse
import itertools
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEGS_OFF = ["--no-cpu-baseline", "--no-e2e", "--no-fe", "--no-cli", "--no-other-workloads", "--project-ranks", "0"]


def pick(d, path):
    for k in path.split("."):
        if d is None:
            return None
        d = d[int(k)] if isinstance(d, list) else d.get(k)
    return d


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--lib", default="")
    ap.add_argument("--workloads", default="c2")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--legs", default="", help="comma list of side legs to keep on: e2e, fe, cli, others, cpu, projection")
    ap.add_argument("--fields", default="ms_per_step,detail.pack_ms_per_step,detail.solve_ms_per_step")
    ap.add_argument("--class-ms-above", type=float, default=0.2, help="also print the per-class solve ms above this (negative: off)")
    ap.add_argument("--tests", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ab"))
    a = ap.parse_args(argv)
    os.makedirs(a.out, exist_ok=True)
    if a.tests:
        r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "-x", "-q", "-k", a.tests], cwd=ROOT, capture_output=True, text=True)
        print("tests rc", r.returncode, r.stdout.strip().splitlines()[-1:] if r.stdout else "")
        if r.returncode != 0:
            print(r.stdout[-3000:])
            return 1
    axes = []
    for e in a.env:
        name, vals = e.split("=", 1)
        axes.append([(name, v) for v in vals.split(",")])
    if a.lib:
        axes.append([("LIB", v) for v in a.lib.split(",")])
    keep = set(filter(None, a.legs.split(",")))
    off = list(LEGS_OFF)
    for leg, flags in (("e2e", ["--no-e2e"]), ("fe", ["--no-fe"]), ("cli", ["--no-cli"]), ("others", ["--no-other-workloads"]), ("cpu", ["--no-cpu-baseline"]),
                       ("projection", ["--project-ranks", "0"])):
        if leg in keep:
            for f in flags:
                off.remove(f)
    fields = a.fields.split(",")
    lib_path = os.path.join(ROOT, "gdmix_amd", "libgdmix_re.so")
    saved_lib = None
    if a.lib:
        saved_lib = tempfile.NamedTemporaryFile(delete=False, suffix=".so").name
        shutil.copy(lib_path, saved_lib)
    best = {}
    try:
        for rep in range(a.reps):
            for combo in itertools.product(*axes) if axes else [()]:
                env = dict(os.environ)
                env.pop("GDMIX_BENCH_LINE", None)
                for name, v in combo:
                    if name == "LIB":
                        shutil.copy(os.path.join(ROOT, v), lib_path)
                        env["GDMIX_ALLOW_STALE_LIB"] = "1"      # a build of other sources under the product's name: the loader refuses those
                    elif v == "unset":
                        env.pop(name, None)
                    else:
                        env[name] = v
                tag = " ".join(f"{n}={v}" for n, v in combo) or "-"
                for w in a.workloads.split(","):
                    detail = os.path.join(a.out, f"{w}_{tag.replace(' ', '_').replace('/', '+')}_{rep}.json")
                    cmd = [sys.executable, "bench.py", "--workload", w, "--steps", str(a.steps), "--warmup", str(a.warmup), "--detail-file", detail] + off + extra
                    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
                    if r.returncode != 0:
                        print(f"{w:12s} {tag}: FAILED rc {r.returncode}: {r.stderr[-400:]}")
                        continue
                    with open(detail) as fh:
                        d = json.load(fh)
                    vals = [pick(d, f) for f in fields]
                    ident = pick(d, "config.library.flags_id")
                    cls = ""
                    if a.class_ms_above >= 0 and pick(d, "detail.class_ms"):
                        cls = "  classes " + str([round(x, 3) for x in d["detail"]["class_ms"] if x > a.class_ms_above])
                    print(f"{w:12s} {tag}: " + "  ".join(f"{f.split('.')[-1]} {v:.4g}" if isinstance(v, (int, float)) else f"{f.split('.')[-1]} {v}"
                                                        for f, v in zip(fields, vals)) + cls + (f"  flags {ident}" if ident else ""), flush=True)
                    if isinstance(vals[0], (int, float)):
                        key = (w, tag)
                        best[key] = min(best.get(key, float("inf")), vals[0])
    finally:
        if saved_lib:
            shutil.copy(saved_lib, lib_path)
            os.unlink(saved_lib)
    print(f"-- best {fields[0]} per variant")
    for (w, tag), v in sorted(best.items()):
        print(f"{w:12s} {tag}: {v:.4g}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
