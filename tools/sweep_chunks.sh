#!/bin/bash
# dev: sweep the number of weight-gradient time chunks (QPN_WGRAD_CHUNKS) on the training bench
for n in 32 48 64 96 128; do
  QPN_WGRAD_CHUNKS=$n timeout -k 10 300 python bench.py --mode train --no-cpu 2>/dev/null > gpurun_out/sw_$n.json
  python - "$n" <<'PY'
import sys, json
n = sys.argv[1]
d = json.loads(open("gpurun_out/sw_%s.json" % n).read().strip().splitlines()[-1]); g = d["roofline"]["groups_ms"]
print(n, round(d["value"], 1), g["k_wgrad"], g["grad_tail"])
PY
done
