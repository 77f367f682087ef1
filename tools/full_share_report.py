#!/usr/bin/env python3
"""bench detail file -> profiles/rNN_c5_full_share.txt (the C5 full-share leg of a default bench run, as text).
    python tools/full_share_report.py gpurun_out/r05ev/bench_detail.json > profiles/r05_c5_full_share.txt"""
import json
import sys

d = json.load(open(sys.argv[1]))["detail"]["c5_full_share"]
print("# round 5 — BASELINE configs[4] at its real per-GPU share, one partition per round (bench_strong.c5_full_share_leg; default bench run of "
      "the evidence session, one MI355X)")
print("# " + d["what"])
head = {k: v for k, v in d.items() if k not in ("what", "round_ms", "projected_rounds", "partitions_per_batch")}
print(json.dumps(head, indent=1))
print("# round_ms of the 128 rounds (one context):")
print(" ".join(str(round(x, 2)) for x in d["round_ms"]))
ppb = d.get("partitions_per_batch")
if ppb:
    print("# " + ppb["what"])
    print("# K = 1 (one context): %.3f s = %.2f M entities/s; three contexts: %.3f s = %.2f M/s" % (
        d["serial_s"], d["serial_entities_per_s"] / 1e6, d["s"], d["entities_per_s"] / 1e6))
    for k in sorted((k for k in ppb if k != "what"), key=int):
        print("# K = %s: %.3f s = %.2f M entities/s (%d converged)" % (k, ppb[k]["s"], ppb[k]["entities_per_s"] / 1e6, ppb[k]["converged"]))
pr = d["projected_rounds"]
print("# " + pr["what"])
print(json.dumps({k: v for k, v in pr.items() if k not in ("what", "rounds")}, indent=1))
print("# round  plain_ms per worker -> slowest (max/mean) | with the plan applied -> slowest (max/mean) | entities moved, wire bytes | priced by")
for i, r in enumerate(pr["rounds"]):
    try:
        print("%3d  %s -> %.2f (%.3f) | %s -> %.2f (%.3f) | %s %s | %s" % (
            i, [round(x, 2) for x in r["plain_ms"]], max(r["plain_ms"]), r["imbalance"], [round(x, 2) for x in r["rebalanced_ms"]],
            max(r["rebalanced_ms"]), r["imbalance_after"], r.get("entities_moved"), int(r.get("wire_bytes_moved") or 0), r.get("priced_by")))
    except Exception:
        print("%3d  %s" % (i, json.dumps(r)[:400]))
