#!/bin/bash
# one-rank RCCL rehearsal of the data-parallel step (process group "nccl", broadcast, all-reduce of the flat buffer, barrier)
export TMPDIR=/tmp
O=gpurun_out
QPN_BENCH_FORCE_PG=1 QPN_EXCHANGE_ALWAYS=1 timeout -k 10 300 python bench.py --mode train --steps 50 --warmup 5 --no-cpu > $O/r2h_nccl1.json 2> $O/r2h_nccl1.err; echo "nccl1 rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2h_nccl1.json").read().strip().splitlines()[-1])
print("one-rank nccl: value %.1f steps/s, backend %s, world_size %s, groups_ms %s" % (d["value"], d["config"]["backend"], d["config"]["world_size"], d["roofline"]["groups_ms"]))
PY
