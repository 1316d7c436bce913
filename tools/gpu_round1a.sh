#!/bin/bash
# first GPU visit: kernel parity + per-layer microbenchmark
mkdir -p gpurun_out
export PYTHONPATH=$PWD
rocminfo | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x --tb=short > gpurun_out/pytest_kernels.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_kernels.log
tail -15 gpurun_out/pytest_kernels.log
timeout 900 python tools/microbench.py --batch 32 --iters 3 --json gpurun_out/microbench.json > gpurun_out/microbench.log 2>&1
echo "microbench rc=$?" >> gpurun_out/microbench.log
tail -40 gpurun_out/microbench.log
