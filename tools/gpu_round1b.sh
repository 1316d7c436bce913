#!/bin/bash
# second GPU visit: all gpu tests, smoke, first bench line, rocprof kernel trace
mkdir -p gpurun_out
export PYTHONPATH=$PWD
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x --tb=short -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|full-size|slowfast_r50_mid|c2d_r50_mid|i3d_r50_mid|Error" gpurun_out/pytest_gpu.log | tail -30
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
ls -R gpurun_out/prof | head -20
