#!/bin/bash
for s in 8 6 5 4 3 2; do
  echo "== QPN_WGRAD_SPLIT=$s"
  QPN_WGRAD_SPLIT=$s python bench.py --mode train --steps 80 --warmup 10 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4))"
done
