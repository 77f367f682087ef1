#!/usr/bin/env python3
"""FETCH_SIZE + WRITE_SIZE (KB, rocprofv3 --pmc, separate passes) per dispatch of the solve kernels ->
profiles/latest_traffic.json keyed by bench.py's class names (bytes per launch)."""
import json
import re
import sqlite3
import sys


def per_dispatch(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, s in cur.execute("select kernel_name, count(*), sum(value) from counters_collection "
                                  "where counter_name = ? group by kernel_name", (counter,)):
        out[name.split("(")[0]] = (n, s)
    return out


def class_name(kernel):
    m = re.search(r"re_solve_grp_kernel<(\d+), (\d+), (\d+), (\d+)>", kernel)
    if m:
        g, epl, ncap, zcap = m.groups()
        return f"re_solve_grp_kernel<{g},{epl}> n<={ncap} nnz<={zcap}"
    return None


fetch, write, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
f = per_dispatch(fetch, "FETCH_SIZE")
w = per_dispatch(write, "WRITE_SIZE")
res = {}
for k, (n, s) in f.items():
    cn = class_name(k)
    if not cn:
        continue
    wn, ws = w.get(k, (0, 0.0))
    # a kernel name is dispatched once per step and per class; classes sharing <G,EPL,NCAP,ZCAP> do not exist
    res[cn] = {"fetch_bytes": s / n * 1024.0, "write_bytes": (ws / wn * 1024.0) if wn else None,
               "bytes": (s / n + (ws / wn if wn else 0.0)) * 1024.0, "dispatches": n,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB), separate passes; 4-byte-per-lane reads, no x2 correction (calibrated, DESIGN.md section 5)"}
json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
