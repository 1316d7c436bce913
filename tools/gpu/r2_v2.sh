#!/bin/bash
# round 2, GPU visit 2: first hardware run of the second-generation implicit GEMM (sf_igemm2.h) -- parity, then A/B per layer
# and end to end (SF_IGEMM2 = 0/1, SF_MATERIALIZE = 0/1).
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q --tb=short -k "igemm2 or conv_fwd or conv_dgrad" > gpurun_out/pytest_i2.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_i2.log | tail -8 | cut -c1-300
timeout 200 python tools/microbench.py --json gpurun_out/microbench_i2.json > gpurun_out/microbench_i2.txt 2>&1; echo "microbench rc=$?"; tail -1 gpurun_out/microbench_i2.txt
for V in "1 1" "0 1" "1 0" "0 0"; do
  set -- $V
  SF_IGEMM2=$1 SF_MATERIALIZE=$2 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_i2_$1_mat_$2.log 2>&1; echo "bench igemm2=$1 materialize=$2 rc=$?"; tail -1 gpurun_out/bench_i2_$1_mat_$2.log | cut -c1-260
done
SF_IGEMM2_MINK=128 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_i2_mink128.log 2>&1; echo "bench mink128 rc=$?"; tail -1 gpurun_out/bench_i2_mink128.log | cut -c1-200
SF_IGEMM2_MINK=1024 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_i2_mink1024.log 2>&1; echo "bench mink1024 rc=$?"; tail -1 gpurun_out/bench_i2_mink1024.log | cut -c1-200
timeout 150 python bench.py --preset MVITv2_S_16x4 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_mvit_i2.log 2>&1; echo "bench mvit rc=$?"; tail -1 gpurun_out/bench_mvit_i2.log | cut -c1-200
SF_IGEMM2_MINK=96 timeout 150 python bench.py --preset MVITv2_S_16x4 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_mvit_i2_mink96.log 2>&1; echo "bench mvit mink96 rc=$?"; tail -1 gpurun_out/bench_mvit_i2_mink96.log | cut -c1-200
