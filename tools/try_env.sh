#!/bin/bash
# dev: train bench under a list of environment settings:  bash tools/try_env.sh "A=1" "B=2 C=3" ...
for e in "" "$@"; do
  echo "== ${e:-default}"
  env $e python bench.py --mode train --steps 80 --warmup 10 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline']['groups_ms'])"
done
