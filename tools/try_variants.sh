#!/bin/bash
# dev: train-step bench with build variants of the library (QPN_LIB)
for so in "" build_variants/*.so; do
  if [ -n "$so" ]; then export QPN_LIB=$PWD/$so; fi
  echo "== ${so:-default}"
  python bench.py --mode train --steps 60 --warmup 10 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline']['groups_ms'])"
done
