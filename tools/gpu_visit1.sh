#!/bin/bash
# GPU visit: kernel parity, model parity, smoke, microbench, bench line, rocprof kernel trace.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
rocminfo | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt 2>&1
nproc >> gpurun_out/device.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short > gpurun_out/pytest_kernels.log 2>&1
echo "pytest kernels rc=$?" | tee -a gpurun_out/pytest_kernels.log
tail -8 gpurun_out/pytest_kernels.log
timeout 600 python tools/microbench.py --batch 32 --iters 3 --json gpurun_out/microbench.json > gpurun_out/microbench.log 2>&1
echo "microbench rc=$?" | tee -a gpurun_out/microbench.log
tail -32 gpurun_out/microbench.log
timeout 900 python -m pytest tests -q -m gpu --tb=short -s --deselect tests/test_kernels_gpu.py > gpurun_out/pytest_gpu.log 2>&1
echo "pytest all rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|error" gpurun_out/pytest_gpu.log | tail -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
ls -R gpurun_out/prof | head -20
