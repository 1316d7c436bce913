#!/bin/bash
# GPU visit 4: full gpu test suite (incl. X3D), MViT bench + rocprof after the rel-pos/softmax fixes, X3D bench + rocprof.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|error|FAILED|x3d" gpurun_out/pytest_gpu.log | tail -25 | cut -c1-700
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log | cut -c1-300
timeout 600 python bench.py --preset MVITv2_S_16x4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_mvit.log 2>&1; echo "bench mvit rc=$?"; tail -1 gpurun_out/bench_mvit.log | cut -c1-2600
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v4_mvit -- python bench.py --preset MVITv2_S_16x4 --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof_mvit.log 2>&1; echo "rocprof mvit rc=$?"
timeout 900 python bench.py --preset X3D_M --batch 64 --steps 5 --warmup 2 > gpurun_out/bench_x3d.log 2>&1; echo "bench x3d rc=$?"; tail -1 gpurun_out/bench_x3d.log | cut -c1-2600
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v4_x3d -- python bench.py --preset X3D_M --batch 64 --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof_x3d.log 2>&1; echo "rocprof x3d rc=$?"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-700
ls gpurun_out/prof | grep v4
