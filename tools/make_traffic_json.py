#!/usr/bin/env python3
"""PMC passes of the bench command (rocprofv3 --pmc, one counter group per pass, tools/profile_round.sh) ->
profiles/latest_traffic.json, keyed by bench.py's size-class names:

    bytes / fetch_bytes / write_bytes   FETCH_SIZE + WRITE_SIZE (KB) per dispatch
    valu_insts                          SQ_INSTS_VALU per dispatch (wave-instructions)
    valu_busy_cycles / wave_cycles      SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES per dispatch (quad-cycles) when collected

    python tools/make_traffic_json.py FETCH.db WRITE.db OUT.json [SQ.db ...] [--workload "text"]
"""
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


args = sys.argv[1:]
workload = None
if "--workload" in args:
    i = args.index("--workload")
    workload = args[i + 1]
    del args[i:i + 2]
fetch, write, out_path, sq = args[0], args[1], args[2], args[3:]
f = per_dispatch(fetch, "FETCH_SIZE")
w = per_dispatch(write, "WRITE_SIZE")
extra = {}
for db in sq:
    for counter, key in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_ACTIVE_INST_VALU", "valu_busy_cycles"), ("SQ_WAVE_CYCLES", "wave_cycles"),
                         ("SQ_BUSY_CYCLES", "sq_busy_cycles")):
        for k, (n, s) in per_dispatch(db, counter).items():
            extra.setdefault(k, {})[key] = s / n
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
    res[cn].update(extra.get(k, {}))
if workload:
    res["_workload"] = workload
# what the counters were collected on: bench.py nulls roofline.traffic when the library it runs is another one
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdmix_amd import build as _build          # noqa: E402
from gdmix_amd import solver as _solver        # noqa: E402
res["_re_abi"] = int(_solver.load_library().gdmix_re_abi_version())
res["_re_kernel_sources_sha16"] = _build.kernel_source_hash()
json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
