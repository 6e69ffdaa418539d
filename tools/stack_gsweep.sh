#!/bin/bash
# dev: backward-queue grid sweep (serial-step kernel average + two-stream step rate per G)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for g in "$@"; do
  QPN_STACK_WGS_BWD=$g QPN_TRAIN_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_g$g -o q --output-format csv -- python3 tools/stack_fwd_time.py 12 > gpurun_out/stack_g$g.log 2>&1 || exit 1
  echo "G=$g $(grep k_stack_bwd gpurun_out/prof_g$g/q_kernel_stats.csv | cut -d, -f4) $(grep 'control words' gpurun_out/stack_g$g.log | cut -c28-100)"
  QPN_STACK_WGS_BWD=$g timeout -k 10 300 python3 tools/stack_rate.py 2>&1 | grep steps
done > gpurun_out/stack_gsweep.txt
