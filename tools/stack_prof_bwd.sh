#!/bin/bash
# dev: kernel trace of a few fused steps, one stream (QPN_TRAIN_SERIAL=1), backward queue on / off
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for bq in 1 0; do
QPN_TRAIN_SERIAL=1 QPN_STACK_QUEUE_BWD=$bq rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bq$bq -o q --output-format csv -- python3 tools/stack_fwd_time.py 20 > gpurun_out/stack_bq$bq.log 2>&1
grep "stack queue" gpurun_out/stack_bq$bq.log
find gpurun_out/prof_bq$bq -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-110 {} | head -14'
done
