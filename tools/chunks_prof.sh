#!/bin/bash
# dev: serial-step kernel averages under QPN_WGRAD_CHUNKS settings
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  QPN_WGRAD_CHUNKS=$n QPN_TRAIN_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ch$n -o q --output-format csv -- python3 tools/stack_fwd_time.py 20 > gpurun_out/ch$n.log 2>&1 || exit 1
done
