"""Randomised parity sweep: the HIP path (through the C ABI) against the CPU oracle on seeded random batches — shapes,
solver options, warm starts and kernel routing drawn at random (tests/fuzz_case.py: one case, the comparison and the adjudication
rule; tests/test_gpu_fuzz.py runs the flagged cases and a 200-case mini-sweep in the GPU suite). This tool is the long sweep.

    PYTHONPATH=.:tests python tools/fuzz_parity.py [cases] [first_seed]
"""
import sys
import time

sys.path.insert(0, "tests")
from fuzz_case import run_case  # noqa: E402
from gdmix_amd.solver import REDeviceSolver  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
solver = REDeviceSolver(0)
bad = adjudicated = 0
worst = 0.0
t0 = time.time()
for case in range(cases):
    r = run_case(solver, seed0 + case)
    worst = max(worst, r["worst_strict_err"])
    adjudicated += len(r["adjudicated"])
    if r["problems"]:
        bad += 1
    if r["problems"] or r["adjudicated"] or case % 10 == 0:
        tag = "BAD" if r["problems"] else ("adj" if r["adjudicated"] else "ok ")
        print(f"{tag} case {r['seed']} {r['shape']:6s} E={r['E']} N={r['N']} Z={r['Z']} strict {r['strict']}/{r['well_posed']} {r['kw']} warm={r['warm']} {r['routing']}"
              + "".join("\n      " + p for p in r["problems"]) + "".join("\n      adjudicated: " + p for p in r["adjudicated"]), flush=True)
print(f"{cases} cases, {bad} with unexplained disagreements, {adjudicated} adjudicated (rounding-level), worst strict theta rel err {worst:.2e}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
