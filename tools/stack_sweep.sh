#!/bin/bash
# dev: fused-step rate with the stack queue at several grid sizes
for g in 512 448 384; do
  echo "QPN_STACK_WGS=$g"; QPN_STACK_WGS=$g python3 tools/stack_rate.py 2>&1 | tail -1
done
echo "per-layer launches"; QPN_STACK_QUEUE=0 python3 tools/stack_rate.py 2>&1 | tail -1
