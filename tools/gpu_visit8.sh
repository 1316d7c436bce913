#!/bin/bash
# GPU visit 8: parity + benches after XCD-aware tile order, unconditional stencil loads, single-stage igemm default.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -12 | cut -c1-600
for P in "X3D_M 64 x3d" "MVITv2_S_16x4 32 mvit" "SLOWFAST_8x8_R50 32 slowfast"; do
  set -- $P
  timeout 600 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3.log 2>&1; echo "bench $3 rc=$?"; tail -1 gpurun_out/bench_$3.log | cut -c1-2000
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v8_$3 -- python bench.py --preset $1 --batch $2 --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof_$3.log 2>&1; echo "rocprof $3 rc=$?"
done
timeout 400 python tools/microbench.py --batch 32 --iters 3 --json gpurun_out/microbench.json > gpurun_out/microbench.log 2>&1; tail -3 gpurun_out/microbench.log
