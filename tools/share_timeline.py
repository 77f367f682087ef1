"""Timeline of the last step in a rocprofv3 --kernel-trace CSV of tools/share_trace.py: start / end of every kernel relative to the
first kernel of the step (the step starts at the last launch of the pack's first kernel).

    python tools/share_timeline.py <dir with *_kernel_trace.csv>
"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
first = names[[i for i, n in enumerate(names) if "pack" in n or "big_" in n][0]]
starts = [i for i, n in enumerate(names) if n == first]
# the last step: from the last "first pack kernel" whose predecessor is not the same kernel
begin = [i for i in starts if i == 0 or names[i - 1] != first][-1]
t0 = int(rows[begin]["Start_Timestamp"])
for r in rows[begin:]:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{a:9.1f} -> {b:9.1f} us  ({b - a:8.1f})  q{r.get('Queue_Id', '?'):>3}  grid {r.get('Grid_Size', '?'):>8} wg {r.get('Workgroup_Size', '?'):>5}  {r['Kernel_Name'][:90]}")
