#!/bin/bash
# round-6 rocprofv3 evidence on the GPU box:  QPN_COMMIT=<hash> bash tools/profile_round5.sh
# kernel-trace stats and the PMC passes are separate runs (FETCH_SIZE / WRITE_SIZE cannot share a pass; no --pmc with trace domains).
TAG=r06
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export QPN_TRAIN_SERIAL=1
TRAIN="python3 bench.py --mode train --steps 20 --warmup 3 --no-cpu"
PMCT="python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu"
DEC="python3 bench.py --mode decode --batch 20 --frames 2005 --steps 1 --warmup 0 --no-cpu"     # the BASELINE workload: 20 x 10 s
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -o train -- $TRAIN > $OUT/train_stats.log 2>&1; echo "train stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode_stats -o decode -- $DEC > $OUT/decode_stats.log 2>&1; echo "decode stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/train_$c -o train -- $PMCT > $OUT/train_$c.log 2>&1; echo "train $c rc=$?"
  rocprofv3 --pmc $c --output-format csv -d $OUT/decode_$c -o decode -- $DEC > $OUT/decode_$c.log 2>&1; echo "decode $c rc=$?"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/train_MFMA -o train -- $PMCT > $OUT/train_MFMA.log 2>&1; echo "train MFMA rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/train_WAIT -o train -- $PMCT > $OUT/train_WAIT.log 2>&1; echo "train WAIT rc=$?"
# the overlapped step (two streams, as the timed loop runs it): kernel stats only
QPN_TRAIN_SERIAL= rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train2_stats -o train -- $TRAIN > $OUT/train2_stats.log 2>&1; echo "train (two streams) stats rc=$?"
python3 tools/profile_summarise.py $TAG
