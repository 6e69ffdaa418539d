#!/bin/bash
# dev: sweep tile height / occupancy cap of the layer kernels
for cfg in "2 0" "2 60000" "2 81000" "1 0" "1 40000" "1 54000" "1 81000"; do
  set -- $cfg
  QPN_LAYER_MT=$1 QPN_LAYER_LDS=$2 timeout -k 10 120 python bench.py --mode train --no-cpu 2>/dev/null > gpurun_out/sl.json
  python - "$1" "$2" <<'PY'
import sys, json
d = json.loads(open("gpurun_out/sl.json").read().strip().splitlines()[-1]); g = d["roofline"]["groups_ms"]
print("MT", sys.argv[1], "lds", sys.argv[2], round(d["value"], 1), g["k_layer_fwd"], g["k_layer_bwd"])
PY
done
