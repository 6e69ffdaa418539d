#!/bin/bash
# dev: where do the waves of the training kernels spend their cycles?  (SQ wait / active counters, one pass)
export TMPDIR=/tmp QPN_TRAIN_SERIAL=1
OUT=gpurun_out/pmc_wait; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o t -- python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu > $OUT/log.txt 2>&1
echo rc=$?
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_wait/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
print("%-42s %10s %8s %8s %8s %8s %8s %8s %8s" % ("kernel", "wave_cyc", "wait", "waitinst", "active", "wlds", "alds", "avmem", "avalu"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:14]:
    w = v["SQ_WAVE_CYCLES"] or 1
    print("%-42s %10.3g %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f" % (k, w, v["SQ_WAIT_ANY"]/w, v["SQ_WAIT_INST_ANY"]/w, v["SQ_ACTIVE_INST_ANY"]/w, v["SQ_WAIT_INST_LDS"]/w, v["SQ_ACTIVE_INST_LDS"]/w, v["SQ_ACTIVE_INST_VMEM"]/w, v["SQ_ACTIVE_INST_VALU"]/w))
PY
