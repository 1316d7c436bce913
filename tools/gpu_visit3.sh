#!/bin/bash
# GPU visit 3: full gpu test suite (incl. MViT), SlowFast bench + rocprof, MViT bench + rocprof.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu --tb=short -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|error|FAILED|mvit" gpurun_out/pytest_gpu.log | tail -25
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 400 python tools/microbench.py --batch 32 --iters 3 --json gpurun_out/microbench.json > gpurun_out/microbench.log 2>&1
echo "microbench rc=$?"; tail -3 gpurun_out/microbench.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-1500
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v3_slowfast -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof_sf.log 2>&1; echo "rocprof slowfast rc=$?"
timeout 900 python bench.py --preset MVITv2_S_16x4 --steps 5 --warmup 2 --no-graph > gpurun_out/bench_mvit.log 2>&1; echo "bench mvit rc=$?"; tail -1 gpurun_out/bench_mvit.log | cut -c1-2500
timeout 600 python bench.py --preset MVITv2_S_16x4 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_mvit_graph.log 2>&1; echo "bench mvit graph rc=$?"; tail -1 gpurun_out/bench_mvit_graph.log | cut -c1-600
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v3_mvit -- python bench.py --preset MVITv2_S_16x4 --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof_mvit.log 2>&1; echo "rocprof mvit rc=$?"
ls gpurun_out/prof | head -20
