#!/usr/bin/env python3
"""tools/inst_mix.sh's table: per kernel, the vector instructions by kind (sums over a kernel's dispatches of two rocprofv3 --pmc passes)."""
import collections
import sqlite3
import sys


def read(db):
    out = collections.defaultdict(dict)
    n = {}
    cur = sqlite3.connect(db).cursor()
    for k, c, cnt, s in cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                                    "where kernel_name like '%gdmix::%' group by kernel_name, counter_name"):
        out[k.split("(")[0]][c] = s
        n[k.split("(")[0]] = cnt
    return out, n


a, na = read(sys.argv[1])
b, _ = read(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] else ({}, {})
rows = sorted(a.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))
print(f"{'kernel':58s} {'launches':>8s} {'VALU (M)':>10s} {'fma64':>6s} {'add64':>6s} {'mul64':>6s} {'trans64':>7s} {'int32':>6s} {'int64':>6s} {'cvt':>5s} {'f32':>5s} "
      f"{'other':>6s} {'lanes/inst':>10s} {'cyc/inst':>8s} {'branch/VALU':>11s}")
for k, v in rows[:16]:
    t = v.get("SQ_INSTS_VALU", 0)
    if t <= 0:
        continue
    w = b.get(k, {})
    pct = lambda x: f"{100.0 * x / t:.1f}"
    named = sum(v.get(c, 0) for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64",
                                      "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"))
    tb = w.get("SQ_INSTS_VALU", 0) or 1
    f32 = sum(w.get(c, 0) for c in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32")) * t / tb
    act = w.get("SQ_ACTIVE_INST_VALU", 0)
    lanes = w.get("SQ_THREAD_CYCLES_VALU", 0) / act if act else float("nan")      # thread-cycles per busy cycle = lanes active, averaged over cycles
    br = w.get("SQ_INSTS_BRANCH", 0) / tb if w else float("nan")
    print(f"{k[-58:]:58s} {na[k]:8d} {t / 1e6:10.1f} {pct(v.get('SQ_INSTS_VALU_FMA_F64', 0)):>6s} {pct(v.get('SQ_INSTS_VALU_ADD_F64', 0)):>6s} "
          f"{pct(v.get('SQ_INSTS_VALU_MUL_F64', 0)):>6s} {pct(v.get('SQ_INSTS_VALU_TRANS_F64', 0)):>7s} {pct(v.get('SQ_INSTS_VALU_INT32', 0)):>6s} "
          f"{pct(v.get('SQ_INSTS_VALU_INT64', 0)):>6s} {pct(v.get('SQ_INSTS_VALU_CVT', 0)):>5s} {pct(f32):>5s} {pct(t - named - f32):>6s} {lanes:10.2f} {br:11.3f}")
print("# columns fma64 ... other: percent of the kernel's SQ_INSTS_VALU (wave-level instructions, summed over its launches); 'other' = what no counter names "
      "(v_mov, DPP moves, compares, selects, permlane, readlane ...). lanes/inst = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU.")
