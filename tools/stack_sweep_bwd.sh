#!/bin/bash
# dev: fused-step rate with the backward stack queue at several grid sizes (the forward queue at its default)
for g in ${SWEEP:-448 416 384 352 320 288}; do
  echo "QPN_STACK_WGS_BWD=$g"; QPN_STACK_WGS_BWD=$g python3 tools/stack_rate.py 2>&1 | tail -1
done
echo "per-layer backward launches"; QPN_STACK_QUEUE_BWD=0 python3 tools/stack_rate.py 2>&1 | tail -1
