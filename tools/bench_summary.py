#!/usr/bin/env python3
"""Print the per-class table of bench.py JSON lines:  python tools/bench_summary.py gpurun_out/r03c/bench_*.json"""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads([ln for ln in open(path) if ln.startswith("{")][0])
    except Exception as e:   # noqa: BLE001
        print(path, "no JSON line:", e)
        continue
    det = d["detail"]
    print(f"== {path}: {d['config'].get('workload_key')}  value {d['value']:.4g} ent/s  ms/step {d['ms_per_step']:.2f} (pack {det['pack_ms_per_step']:.2f} "
          f"solve {det['solve_ms_per_step']:.2f})  nit {det['mean_nit']:.2f} nfev {det['mean_nfev']:.2f}  W/D {det['parity_classes']}  n_gpus {d['n_gpus']}")
    for c in det.get("per_class", []):
        print(f"   {c['kernel']:50s} E={c['entities']:8d} ms={c['ms']:8.3f} alg {c['alg_GBps']:7.1f} restream {c['restreamed_GBps']:7.1f} GB/s "
              f"n={c['mean_n']:8.1f} p={c['mean_p']:7.1f} maxnnz={c['max_nnz']} nfev {c['mean_nfev']}")
    r = d["roofline"]
    print(f"   roofline: {r['kernel']}  {r['achieved']:.1f} GB/s = {r['frac']:.4f} of peak; launch {r['avg_launch_ms']:.3f} ms")
    if d.get("cpu_baseline"):
        print("   cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:90])
