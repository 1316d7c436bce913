#!/bin/bash
# GPU visit 6: gpu tests (incl. Nonlocal), benches for the three headline models, PMC traffic passes for SlowFast.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -12 | cut -c1-600
for P in "MVITv2_S_16x4 32 mvit" "X3D_M 64 x3d" "SLOWFAST_8x8_R50 32 slowfast"; do
  set -- $P
  timeout 600 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3.log 2>&1; echo "bench $3 rc=$?"; tail -1 gpurun_out/bench_$3.log | cut -c1-1800
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v6_mvit -- python bench.py --preset MVITv2_S_16x4 --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof_mvit.log 2>&1; echo "rocprof mvit rc=$?"
# PMC passes (counters alone, no stats): eager mode so that every dispatch is a separately profiled kernel
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc -o fetch -- python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-profile > gpurun_out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc -o write -- python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-profile > gpurun_out/pmc_write.log 2>&1; echo "pmc write rc=$?"
ls gpurun_out/pmc gpurun_out/prof | grep -E "v6|fetch|write"
python tools/pmc_traffic.py gpurun_out/pmc/fetch_counter_collection.csv gpurun_out/pmc/write_counter_collection.csv gpurun_out/pmc_traffic.json 2>&1 | tail -14
