#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd .db outputs (kernel-trace stats and PMC passes) as plain text.

    python tools/prof_summary.py --stats gpurun_out/prof/x_results.db [--pmc a.db b.db ...] > profiles/rNN_xxx.txt
"""
import argparse
import sqlite3


def short(name, n=70):
    name = name.split("(")[0]
    return name if len(name) <= n else name[:n - 3] + "..."


def stats(db):
    cur = sqlite3.connect(db).cursor()
    print(f"# kernel-trace stats: {db}")
    print(f"{'kernel':72s} {'calls':>6s} {'total_us':>12s} {'avg_us':>11s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                            "from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    for name, calls, total, avg, mn, mx in rows[:25]:
        print(f"{short(name):72s} {calls:6d} {total / 1e3:12.1f} {avg / 1e3:11.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * total / tot:6.2f}")
    print("# launch geometry / resources of the gdmix kernels, by total time (first dispatch of each; template kernels are named `void gdmix::...`)")
    for r in cur.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count "
                         "from kernels where name like '%gdmix::%' group by name order by sum(duration) desc"):
        print(f"{short(r[0]):60s} grid={r[1]} wg={r[2]} lds={r[3]} scratch={r[4]} vgpr={r[5]} agpr={r[6]} sgpr={r[7]}")


def pmc(db):
    cur = sqlite3.connect(db).cursor()
    print(f"# PMC pass: {db}")
    rows = list(cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                            "where kernel_name like '%gdmix::%' group by kernel_name, counter_name order by sum(value) desc"))
    print(f"{'kernel':60s} {'counter':18s} {'dispatches':>10s} {'sum':>16s} {'avg/dispatch':>16s}")
    for k, c, n, s, a in rows:
        print(f"{short(k, 60):60s} {c:18s} {n:10d} {s:16.1f} {a:16.2f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--pmc", nargs="*", default=[])
    a = ap.parse_args()
    if a.stats:
        stats(a.stats)
    for p in a.pmc:
        pmc(p)
