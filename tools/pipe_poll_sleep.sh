#!/bin/bash
# dev: the pipelined decode kernel (decode_pipe.hip) with an s_sleep between two polls of a wait -- build the variants first:
#   for n in 1 2 4 8; do python qpnet_amd/csrc/build.py --variant pollsleep$n -DPIPE_POLL_SLEEP=$n; done
# then on the GPU box: bash tools/pipe_poll_sleep.sh   (bench.py's decode leg: 20 x 10 s utterances, and B = 1)
for n in 0 1 2 4 8; do
  if [ $n = 0 ]; then unset QPN_LIB; else export QPN_LIB=$PWD/build_variants/libqpnet_pollsleep$n.so; fi
  for B in 20 1; do
    python3 bench.py --mode decode --no-cpu --batch $B --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sleep %s  B=%-2d  %.3f M samples/s  %.2f us per sample per utterance' % ('$n', $B, d['value']/1e6, d['roofline']['achieved']))" || exit 1
  done
done
