#!/bin/bash
# dev: kernel-trace stats + SQ wait counters + matrix-core busy cycles of the paper-size training step (three separate runs)
export TMPDIR=/tmp QPN_TRAIN_SERIAL=1
OUT=gpurun_out/pmc_r3; rm -rf $OUT; mkdir -p $OUT
TRAIN="python3 bench.py --mode train --steps 20 --warmup 3 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o t -- $TRAIN > $OUT/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/wait -o t -- python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu > $OUT/wait.log 2>&1; echo "wait rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o t -- python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu > $OUT/mfma.log 2>&1; echo "mfma rc=$?"
python3 - <<'PY'
import csv, glob, collections
st = glob.glob("gpurun_out/pmc_r3/stats/**/*kernel_stats.csv", recursive=True)
avg = {}
if st:
    rows = list(csv.DictReader(open(st[0])))
    print("%-60s %6s %10s %6s" % ("kernel", "calls", "avg_us", "%"))
    for r in rows[:22]:
        avg[r["Name"].split("(")[0][:60]] = float(r["AverageNs"])
        print("%-60s %6s %10.1f %6s" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
f = glob.glob("gpurun_out/pmc_r3/wait/**/*counter_collection.csv", recursive=True)
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
    print("%-42s %10s %8s %8s %8s %8s %8s %8s %8s" % ("kernel", "wave_cyc", "wait", "waitinst", "active", "wlds", "alds", "avmem", "avalu"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:16]:
        w = v["SQ_WAVE_CYCLES"] or 1
        print("%-42s %10.3g %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f" % (k[:42], w, v["SQ_WAIT_ANY"]/w, v["SQ_WAIT_INST_ANY"]/w, v["SQ_ACTIVE_INST_ANY"]/w, v["SQ_WAIT_INST_LDS"]/w, v["SQ_ACTIVE_INST_LDS"]/w, v["SQ_ACTIVE_INST_VMEM"]/w, v["SQ_ACTIVE_INST_VALU"]/w))
f = glob.glob("gpurun_out/pmc_r3/mfma/**/*counter_collection.csv", recursive=True)
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES": n[k] += 1
    print("%-42s %6s %14s %10s %12s" % ("kernel", "calls", "mfma_busy/call", "mfma_util", "clk_GHz(est)"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"])[:16]:
        c = max(n[k], 1); ns = avg.get(k)
        util = v["SQ_VALU_MFMA_BUSY_CYCLES"] / c / (ns * 2.4 * 1024) if ns else float("nan")
        clk = v["GRBM_GUI_ACTIVE"] / c / 8 / ns if ns else float("nan")
        print("%-42s %6d %14.4g %10.3f %12.2f" % (k[:42], c, v["SQ_VALU_MFMA_BUSY_CYCLES"] / c, util, clk))
PY
