"""Per-kernel duration by step bucket from a rocprofv3 --kernel-trace of tools/warmup_curve.py: which kernels are slower in a process's first steps?
    python tools/warm_trace_summary.py <dir with *_kernel_trace.csv> [steps]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
by = defaultdict(list)
for s, e, n in rows:
    by[n].append((s, e - s))
print("%d dispatches, %d kernels" % (len(rows), len(by)))
for n, v in sorted(by.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    if len(v) < 40:
        continue
    per = len(v) // 80 if len(v) >= 80 else 1          # launches per step (80 steps)
    d = [x[1] / 1e3 for x in v]
    k = len(d)
    b = [d[int(k * a):int(k * b_)] for a, b_ in ((0.02, 0.1), (0.1, 0.2), (0.2, 0.3), (0.3, 0.5), (0.5, 0.75), (0.75, 1.0))]
    print("%-60s n=%4d  us by run fraction [2-10%% 10-20%% 20-30%% 30-50%% 50-75%% 75-100%%]: %s" % (n[:60], k, " ".join("%.1f" % (sum(x) / max(len(x), 1)) for x in b)))
