#!/bin/bash
# GPU visit 18: full validation -- gpu suite, smoke, default bench (with cpu_baseline), the other presets, rocprof of the
# default bench command.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -8 | cut -c1-600
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
/usr/bin/time -v timeout 900 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "bench default rc=$?"; tail -1 gpurun_out/bench_default.log | cut -c1-3000; grep -E "Elapsed" gpurun_out/bench_default.err
for P in "C2D_8x8_R50 32 c2d" "X3D_M 64 x3d" "MVITv2_S_16x4 32 mvit"; do
  set -- $P
  timeout 600 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3.log 2>&1; echo "bench $3 rc=$?"; tail -1 gpurun_out/bench_$3.log | cut -c1-400
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v18_slowfast -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof_slowfast.log 2>&1; echo "rocprof slowfast rc=$?"
