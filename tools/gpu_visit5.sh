#!/bin/bash
# GPU visit 5: gpu tests, then MViT / X3D / SlowFast benches with rocprof after the blocked depthwise kernels and rel-pos rewrite
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -12 | cut -c1-400
for P in "MVITv2_S_16x4 32 mvit" "X3D_M 64 x3d" "SLOWFAST_8x8_R50 32 slowfast"; do
  set -- $P
  timeout 600 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3.log 2>&1; echo "bench $3 rc=$?"; tail -1 gpurun_out/bench_$3.log | cut -c1-2400
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v5_$3 -- python bench.py --preset $1 --batch $2 --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof_$3.log 2>&1; echo "rocprof $3 rc=$?"
done
ls gpurun_out/prof | grep v5
