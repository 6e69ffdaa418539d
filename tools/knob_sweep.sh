#!/bin/bash
# dev: two-stream step rate under a list of environment settings ("NAME=VALUE[,NAME=VALUE]" per argument; "-" = none)
#   bash tools/knob_sweep.sh - QPN_STACK_WGS_BWD=512 QPN_STACK_WGS_BWD=256,QPN_WGRAD_CHUNKS_SIDE=32
for spec in "$@"; do
  envs=""
  if [ "$spec" != "-" ]; then envs=$(echo "$spec" | tr ',' ' '); fi
  v=$(env $envs python3 bench.py --mode train --steps 300 --warmup 30 --no-cpu 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))")
  echo "$spec : $v"
done
