#!/bin/bash
# round-3 rocprofv3 evidence on the GPU box:  bash tools/profile_round3.sh
# kernel-trace stats and the PMC passes are separate runs (FETCH_SIZE / WRITE_SIZE cannot share a pass).
TAG=r03
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
export QPN_TRAIN_SERIAL=1
TRAIN="python3 bench.py --mode train --steps 20 --warmup 3 --no-cpu"
DEC="python3 bench.py --mode decode --batch 20 --frames 2005 --steps 1 --warmup 0 --no-cpu"     # the BASELINE workload: 20 x 10 s
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -o train -- $TRAIN > $OUT/train_stats.log 2>&1; echo "train stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode_stats -o decode -- $DEC > $OUT/decode_stats.log 2>&1; echo "decode stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/train_$c -o train -- python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu > $OUT/train_$c.log 2>&1; echo "train $c rc=$?"
  rocprofv3 --pmc $c --output-format csv -d $OUT/decode_$c -o decode -- $DEC > $OUT/decode_$c.log 2>&1; echo "decode $c rc=$?"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/train_MFMA -o train -- python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu > $OUT/train_MFMA.log 2>&1; echo "train MFMA rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/train_WAIT -o train -- python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu > $OUT/train_WAIT.log 2>&1; echo "train WAIT rc=$?"
# the repo-default geometry: kernel stats of a few training steps + a short cooperative decode
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default_stats -o default -- python3 tools/bench_default.py > $OUT/default_stats.log 2>&1; echo "default train stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default_dec_stats -o defdec -- python3 tools/bench_decode_coop.py default500 > $OUT/default_dec_stats.log 2>&1; echo "default decode stats rc=$?"
python3 tools/profile_summarise.py $TAG
