# dev: where one overlapped training step spends its wall time -- from a rocprofv3 --kernel-trace csv of `bench.py --mode train`:
#   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --mode train --steps 12 --warmup 3 --no-cpu
#   python3 tools/step_timeline.py gpurun_out/tl
# prints, for the median of the last steps, every kernel's start offset from the step's first kernel, its duration, queue, and the idle gaps of the main queue
import csv, glob, sys, re, statistics
root = sys.argv[1]
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def nm(r):
    n = r["Kernel_Name"]
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:44]
ks = sorted(({"n": nm(r), "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]), "q": r.get("Queue_Id", "?")} for r in rows), key=lambda k: k["s"])
starts = [i for i, k in enumerate(ks) if k["n"].startswith("k_refresh")]
steps = [ks[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = [s for s in steps if any(k["n"].startswith("k_adam") for k in s)][-8:]
durs = [s[-1]["e"] - s[0]["s"] for s in steps]
period = [b[0]["s"] - a[0]["s"] for a, b in zip(steps[:-1], steps[1:])]
inter = [b[0]["s"] - max(k["e"] for k in a) for a, b in zip(steps[:-1], steps[1:])]
allsteps = [ks[a:b] for a, b in zip(starts[:-1], starts[1:])]
allinter = sorted((b[0]["s"] - max(k["e"] for k in a)) / 1e3 for a, b in zip(allsteps[:-1], allsteps[1:]) if a and b)
if allinter:
    n = len(allinter)
    print("idle between a step's last kernel and the next step's first, all %d steps: median %.1f us, mean %.1f, p90 %.1f, max %.1f" % (n, allinter[n // 2], sum(allinter) / n, allinter[int(n * 0.9)], allinter[-1]))
print("steps analysed %d; first-kernel-start to last-kernel-end: median %.1f us; step period median %.1f us" % (len(steps), statistics.median(durs) / 1e3, statistics.median(period) / 1e3 if period else 0))
s = steps[len(steps) // 2]
t0 = s[0]["s"]
mainq = s[0]["q"]
prev_end = t0
print("%-46s %5s %9s %9s %9s" % ("kernel", "queue", "start us", "dur us", "gap us"))
for k in s:
    gap = ""
    if k["q"] == mainq:
        gap = "%.1f" % ((k["s"] - prev_end) / 1e3)
        prev_end = max(prev_end, k["e"])
    print("%-46s %5s %9.1f %9.1f %9s" % (k["n"], "main" if k["q"] == mainq else "side", (k["s"] - t0) / 1e3, (k["e"] - k["s"]) / 1e3, gap))
