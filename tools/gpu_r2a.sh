#!/bin/bash
# round-2 first GPU pass: parity tests, default bench, the multi-rank rehearsals
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r2a_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r2a_pytest.log
python bench.py --steps 30 --warmup 5 > $O/r2a_bench.json 2> $O/r2a_bench.err; echo "bench rc=$?"
QPN_BENCH_ONE_GPU=1 python bench.py --gpus 2 --mode train --steps 30 --warmup 5 --no-cpu > $O/r2a_gloo2.json 2> $O/r2a_gloo2.err; echo "gloo2 rc=$?"
QPN_BENCH_FORCE_PG=1 QPN_EXCHANGE_ALWAYS=1 python bench.py --mode train --steps 30 --warmup 5 --no-cpu > $O/r2a_nccl1.json 2> $O/r2a_nccl1.err; echo "nccl1 rc=$?"
QPN_BENCH_ONE_GPU=1 QPN_DIST_BACKEND=nccl timeout -k 10 120 python bench.py --gpus 2 --mode train --steps 10 --warmup 2 --no-cpu > $O/r2a_nccl2.json 2> $O/r2a_nccl2.err; echo "nccl2-on-one-gpu rc=$? (expected to be refused by RCCL: duplicate device)"
true
