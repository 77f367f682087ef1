"""detail.strong_projection of a bench.py JSON line as text (profiles/r04_strong_projection.txt).   python tools/projection_report.py bench.json"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
det = d["detail"]
for p in det["strong_projection"] or []:
    print(f"== {p['workload']}: {p.get('what', '')}")
    print(f"   total {p['total_entities']} entities, {p.get('total_nnz', '?')} non-zeros, {p.get('partitions', '?')} partitions; job = slowest share {p['ms']:.2f} ms -> "
          f"{p['entities_per_s'] / 1e6:.2f} M entities/s; imbalance max/mean {p['imbalance']:.3f}; sum of shares {sum(r['ms_per_step'] for r in p['per_rank']):.1f} ms")
    for r in p["per_rank"]:
        top = ", ".join(f"{k['kernel'].replace('re_solve_', '')} {k['entities']} ent {k['ms']} ms" for k in r.get("largest_launches", [])[:4])
        print(f"   share {r['rank']}: {r['entities']:>8} entities {r['nnz']:>11} nnz  {r['ms_per_step']:8.2f} ms  (pack {r.get('pack_ms', float('nan')):.2f}, class launches "
              f"{r.get('solve_kernel_ms', float('nan')):.2f})  largest entity {r.get('largest_nnz', '?')} nnz | {top}")
    rp = p.get("rebalance_plan") or {}
    if rp:
        print(f"   re-balancing plan on measured costs (tolerance {rp.get('tolerance')}): imbalance {rp.get('predicted_imbalance')} -> {rp.get('after_imbalance')} "
              f"by moving {rp.get('entities_to_move')} entities = {(rp.get('wire_bytes_to_move') or 0) / 1e6:.1f} MB of wire form (plan only)")
    pr = p.get("partition_rounds")
    if pr:
        print(f"   per-partition rounds (the product path's granularity): mean imbalance {pr.get('mean_imbalance')}, worst {pr.get('worst_imbalance')}")
    print()
w = det.get("workloads") or {}
print("# the same run's single-GPU legs over the WHOLE populations (detail.workloads): " +
      "; ".join(f"{k} {v['ms_per_step']:.2f} ms = {v['entities_per_s'] / 1e6:.2f} M entities/s" for k, v in w.items() if "ms_per_step" in v))
print(f"# headline: C2 {d['ms_per_step']:.2f} ms per step = {d['value'] / 1e6:.1f} M entities/s ({d['steps']} steps, {d['warmup']} warm-up)")
