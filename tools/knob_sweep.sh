#!/bin/bash
# dev: two-stream step rate under a list of environment settings ("NAME=VALUE[,NAME=VALUE]" per argument; "-" = none)
cd $GRAFT_REPO_ROOT
for kv in "$@"; do
  envs=""; if [ "$kv" != "-" ]; then envs=$(echo $kv | tr ',' ' '); fi
  r=$(env $envs timeout -k 10 300 python3 tools/stack_rate.py 2>&1 | grep steps | cut -c1-40)
  echo "$kv: $r"
done > gpurun_out/knob_sweep.txt
