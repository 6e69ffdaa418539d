#!/bin/bash
# dev: serial-step kernel averages of experiment builds (build_variants/libqpnet_exp*.so: parts of the backward queue's memory traffic removed -- timing only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for f in build_variants/libqpnet_exp*.so; do
  v=$(basename $f .so)
  QPN_LIB=$f QPN_TRAIN_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$v -o q --output-format csv -- python3 tools/stack_fwd_time.py 12 > gpurun_out/stack_$v.log 2>&1 || exit 1
  echo "$v: $(grep k_stack gpurun_out/prof_$v/q_kernel_stats.csv | cut -d, -f1,4 | cut -c1-30,60- | tr '\n' ' ')"
done > gpurun_out/stack_exp.txt
