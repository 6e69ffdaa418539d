#!/bin/bash
# dev: SQ wait / LDS counters of the wide-stack GEMM kernels (C=512 training step)
export TMPDIR=/tmp QPN_TRAIN_SERIAL=1
OUT=gpurun_out/pmc_gemm; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -o t -- python3 tools/bench_default.py > $OUT/log_a.txt 2>&1
echo rc=$?
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/b -o t -- python3 tools/bench_default.py > $OUT/log_b.txt 2>&1
echo rc=$?
python3 - <<'PY'
import csv, glob, collections
for sub in ("a", "b"):
    f = glob.glob("gpurun_out/pmc_gemm/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not f: print("no csv for", sub); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][:34]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    names = sorted({c for v in agg.values() for c in v})
    print("%-36s" % "kernel", " ".join("%14s" % n[-14:] for n in names))
    key = "SQ_WAVE_CYCLES" if sub == "a" else "SQ_BUSY_CYCLES"
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][key])[:10]:
        print("%-36s" % k, " ".join("%14.4g" % v[n] for n in names))
PY
