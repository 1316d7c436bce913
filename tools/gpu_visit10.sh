#!/bin/bash
# GPU visit 10: direct-to-LDS GEMM operands (GL), vectorised wgrad reduce, register-staged loads restored,
# SlowFast-R101+Nonlocal preset (BASELINE config 5 backbone).
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -12 | cut -c1-600
for P in "SLOWFAST_8x8_R50 32 slowfast" "MVITv2_S_16x4 32 mvit" "X3D_M 64 x3d"; do
  set -- $P
  timeout 600 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3.log 2>&1; echo "bench $3 rc=$?"; tail -1 gpurun_out/bench_$3.log | cut -c1-1500
  SF_IGEMM_GLDS=0 timeout 600 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_$3_noglds.log 2>&1; echo "bench $3 (no glds) rc=$?"; tail -1 gpurun_out/bench_$3_noglds.log | cut -c1-400
done
timeout 900 python bench.py --preset SLOWFAST_32x2_R101_50_50 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r101nl.log 2>&1; echo "bench r101nl rc=$?"; tail -1 gpurun_out/bench_r101nl.log | cut -c1-1500
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v10_mvit -- python bench.py --preset MVITv2_S_16x4 --batch 32 --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof_mvit.log 2>&1; echo "rocprof mvit rc=$?"
