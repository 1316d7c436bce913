#!/bin/bash
# GPU visit 20: vectorised BN+ReLU operand transform in wgrad too; benches of the three headline models.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --tb=short -k "wgrad or blocks or model_matches" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu (subset) rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -8 | cut -c1-600
for P in "SLOWFAST_8x8_R50 32 slowfast" "X3D_M 64 x3d" "MVITv2_S_16x4 32 mvit"; do
  set -- $P
  timeout 600 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3.log 2>&1; echo "bench $3 rc=$?"; tail -1 gpurun_out/bench_$3.log | cut -c1-1500
done
