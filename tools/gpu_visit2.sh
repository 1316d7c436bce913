#!/bin/bash
# GPU visit 2: parity (kernels, blocks, models, graph step), microbench, bench (graph + eager), rocprof.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --tb=short -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|error|FAILED" gpurun_out/pytest_gpu.log | tail -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 400 python tools/microbench.py --batch 32 --iters 3 --json gpurun_out/microbench.json > gpurun_out/microbench.log 2>&1
echo "microbench rc=$?"; tail -30 gpurun_out/microbench.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -2 gpurun_out/bench.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_eager.log 2>&1; echo "bench eager rc=$?"; tail -1 gpurun_out/bench_eager.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v2 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
ls -R gpurun_out/prof | head -20
