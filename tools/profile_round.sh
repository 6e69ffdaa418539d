#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box:  bash tools/profile_round.sh <tag>   (e.g. r01)
# kernel-trace stats and the two PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass) are separate runs.
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
# one stream, as in the per-group roofline measurement of bench.py (the timed loop overlaps the skip/post weight gradients with
# the layer backward on a side stream, which stretches the individual kernel durations)
export QPN_TRAIN_SERIAL=1
TRAIN="python3 bench.py --mode train --steps 20 --warmup 3 --no-cpu"
DEC="python3 bench.py --mode decode --batch 20 --frames 600 --steps 1 --warmup 0 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -o train -- $TRAIN > $OUT/train_stats.log 2>&1
echo "train stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode_stats -o decode -- $DEC > $OUT/decode_stats.log 2>&1
echo "decode stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/train_$c -o train -- python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu > $OUT/train_$c.log 2>&1
  echo "train $c done"
  rocprofv3 --pmc $c --output-format csv -d $OUT/decode_$c -o decode -- $DEC > $OUT/decode_$c.log 2>&1
  echo "decode $c done"
done
# matrix-core busy cycles per kernel (MFMA utilisation = busy / (duration x 2.4 GHz x 1024 SIMDs))
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/train_MFMA -o train -- python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu > $OUT/train_MFMA.log 2>&1
echo "train MFMA done"
python3 tools/profile_summarise.py $TAG
