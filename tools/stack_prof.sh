#!/bin/bash
# dev: kernel trace of a few fused steps with the stack queue at two grid sizes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for g in 512 256; do
QPN_STACK_WGS=$g rocprofv3 --kernel-trace --stats -d gpurun_out/prof_q$g -o q --output-format csv -- python3 tools/stack_fwd_time.py 20 > gpurun_out/stack_q$g.log 2>&1
grep "stack queue" gpurun_out/stack_q$g.log
find gpurun_out/prof_q$g -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-120 {} | grep -i "stack\|layer_bwd_p<11, false\|up_bwd"'
done
