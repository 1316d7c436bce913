#!/bin/bash
# GPU visit 23: final validation of the round -- full gpu suite, smoke, default bench (cpu_baseline included), other presets.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -8 | cut -c1-600
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
SECONDS=0
timeout 900 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "bench default rc=$? wall=${SECONDS}s"; tail -1 gpurun_out/bench_default.log | cut -c1-700
for P in "MVITv2_S_16x4 32 mvit" "X3D_M 64 x3d" "SLOWFAST_32x2_R101_50_50 16 ava"; do
  set -- $P
  timeout 600 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3.log 2>&1; echo "bench $3 rc=$?"; tail -1 gpurun_out/bench_$3.log | cut -c1-330
done
