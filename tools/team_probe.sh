#!/bin/bash
GDMIX_EXTRA_FLAGS=-DGDMIX_TEAM_PROFILE python -m gdmix_amd.build --force > /dev/null 2>&1
PYTHONPATH=. python tools/team_probe.py ${1:-580} ${2:-512} > /tmp/probe.log 2>&1
grep "^ms" /tmp/probe.log
python - <<'PY'
import re, numpy as np
rows = []
for l in open("/tmp/probe.log"):
    if l.startswith("team ") and l.rstrip().count(" ") == 18:
        m = re.findall(r"[a-z0-9]+ ([0-9.]+)", l.split("us/eval:")[1])
        nf = int(re.search(r"nfev=(\d+)", l).group(1))
        rows.append([nf] + [float(x) for x in m])
a = np.array(rows)
print("entities", len(a), "median nfev %.0f" % np.median(a[:, 0]))
names = ["rows", "red1", "cols", "red2", "solve", "upd", "sync"]
for q in (10, 50, 90):
    v = np.percentile(a[:, 1:], q, axis=0)
    print("p%d us/eval: " % q + " ".join("%s %.1f" % (n, x) for n, x in zip(names, v)) + "  total %.1f" % v.sum())
PY
