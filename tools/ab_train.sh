#!/bin/bash
# A/B of training-step variants on one box: tools/ab_train.sh "NAME=ENV..." ...   (each arg: label=ENVVAR=val,ENVVAR=val or label=)
# prints steps/s and the per-group device times of each variant (bench.py --mode train --no-cpu)
for spec in "$@"; do
  label="${spec%%=*}"; envs="${spec#*=}"
  ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
    python bench.py --mode train --no-cpu > "gpurun_out/ab_${label}.json" 2> "gpurun_out/ab_${label}.err" )
  python - "$label" <<'PY'
import json, sys
lab = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/ab_%s.json" % lab).read().strip().splitlines()[-1])
    print(lab, round(d["value"], 1), "steps/s", d["roofline"]["groups_ms"])
except Exception as e:
    print(lab, "FAILED", e, open("gpurun_out/ab_%s.err" % lab).read()[-400:])
PY
done
