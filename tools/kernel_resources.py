#!/usr/bin/env python3
"""Static resources of every kernel in libgdmix_re.so, read from the code objects' AMDGPU metadata (no GPU needed):
VGPRs, AGPRs, SGPRs, spilled registers, scratch bytes per lane, static LDS, maximum workgroup size.

    python tools/kernel_resources.py [--lib gdmix_amd/libgdmix_re.so] [--match re_solve] > profiles/rNN_kernel_resources.txt

What the rocprofv3 summaries (tools/prof_summary.py) show per dispatch, this shows for every instantiation, including those a workload
never launches — and the spill counts, which the dispatch record does not carry. Dynamic LDS (the solvers' arenas) is a launch
argument: the dispatch table of the profile has it.
"""
import argparse
import os
import re
import shutil
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = (".name", ".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
          ".group_segment_fixed_size", ".max_flat_workgroup_size")


def demangle(names):
    p = subprocess.run([shutil.which("c++filt") or f"{LLVM}/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines() if p.returncode == 0 else names


def kernel_symbols(lib):
    """name -> (vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, max_wg) from the metadata's kernel records."""
    recs = []
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", copy], check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            # a kernel record is the run of 4-space-indented keys between two "  - .agpr_count:" lines; argument records are deeper
            for block in re.split(r"\n  - (?=\.agpr_count:)", notes)[1:]:
                rec = {}
                for line in block.splitlines():
                    m = re.match(r"(?:    )?(\.\w+):\s+(\S.*)$", line)
                    if m and m.group(1) in FIELDS and m.group(1) not in rec:
                        rec[m.group(1)] = m.group(2).strip().strip("'")
                if ".name" in rec and ".vgpr_count" in rec:
                    recs.append(rec)
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "gdmix_amd", "libgdmix_re.so"))
    ap.add_argument("--match", default="", help="only kernels whose demangled name contains this")
    a = ap.parse_args()
    recs = kernel_symbols(a.lib)
    names = demangle([r[".name"] for r in recs])
    rows = []
    for r, n in zip(recs, names):
        n = re.sub(r"^void ", "", n.split("(")[0])
        if a.match and a.match not in n:
            continue
        rows.append((n, *(int(r.get(k, 0)) for k in FIELDS[1:])))
    rows.sort()
    try:
        import sys
        sys.path.insert(0, ROOT)
        from gdmix_amd import build
        stamp = f"build id {build.embedded_id(a.lib)} (sources hash to {build.source_id()}), flags {build.embedded_id(a.lib, build.FLAGS_MARKER)}"
    except Exception as e:  # noqa: BLE001
        stamp = f"(no build id: {e})"
    print(f"# static kernel resources of {os.path.relpath(a.lib, ROOT)} — {stamp}")
    print("# vgpr/agpr/sgpr = allocated registers; vspill/sspill = spilled registers; scratch = bytes per lane; lds = static bytes (dynamic LDS is a launch argument)")
    print(f"{'kernel':78s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>8s} {'lds':>7s} {'max_wg':>6s}")
    for n, v, ag, s, vs, ss, scr, lds, wg in rows:
        print(f"{n[:78]:78s} {v:5d} {ag:5d} {s:5d} {vs:6d} {ss:6d} {scr:8d} {lds:7d} {wg:6d}")
    print(f"# {len(rows)} kernels")


if __name__ == "__main__":
    main()
