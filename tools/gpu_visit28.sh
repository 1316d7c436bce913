#!/bin/bash
# GPU visit 28 (last of the round): kernel / token suites after the igemm epilogue change (mode 3) + X3D-M training bench.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 75 python -m pytest tests/test_kernels_gpu.py tests/test_tokens_gpu.py -q --tb=short -x > gpurun_out/pytest_gpu28.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu28.log | tail -6 | cut -c1-400
timeout 40 python bench.py --preset X3D_M --batch 64 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_x3d28.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_x3d28.log | cut -c1-300
