#!/bin/bash
# GPU visit 21: attention with pre-scaled q/k + chained bias MFMAs, packed fp16 ReLU in the operand transform: full suite + benches.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -8 | cut -c1-600
for P in "SLOWFAST_8x8_R50 32 slowfast" "MVITv2_S_16x4 32 mvit" "X3D_M 64 x3d"; do
  set -- $P
  timeout 600 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3.log 2>&1; echo "bench $3 rc=$?"; tail -1 gpurun_out/bench_$3.log | cut -c1-1200
done
