#!/bin/bash
# dev: one iteration of the stack-queue tuning loop on the GPU box: tile stamps of the instrumented build (build_variants/libqpnet_stamp*.so),
# kernel averages of the serial step, the two-stream step rate
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in stampb stampf; do
  if [ -f build_variants/libqpnet_$v.so ]; then
    QPN_LIB=build_variants/libqpnet_$v.so QPN_TRAIN_SERIAL=1 timeout -k 10 200 python3 tools/stack_fwd_time.py 6 > gpurun_out/stack_$v.log 2>&1 || exit 1
  fi
done
QPN_TRAIN_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_iter -o q --output-format csv -- python3 tools/stack_fwd_time.py 20 > gpurun_out/stack_iter_prof.log 2>&1 || exit 1
find gpurun_out/prof_iter -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-100 {} | head -8' > gpurun_out/stack_iter_stats.txt
timeout -k 10 300 python3 tools/stack_rate.py > gpurun_out/stack_iter_rate.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/stack_rate.py >> gpurun_out/stack_iter_rate.log 2>&1
if [ -f build_variants/libqpnet_noprio.so ]; then QPN_LIB=build_variants/libqpnet_noprio.so timeout -k 10 300 python3 tools/stack_rate.py > gpurun_out/stack_iter_rate_noprio.log 2>&1; fi
