#!/bin/bash
# GPU visit 11: DropPath (sf_row_scale_add, ABI v8), faster depthwise wgrad finalize / strided dgrad, PMC traffic refresh
# for SlowFast after the XCD-aware tile order, R101+NL at batch 16.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short -k "drop_path or mvit or token or dwconv or x3d or abi" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu (subset) rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -12 | cut -c1-600
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
for P in "MVITv2_S_16x4 32 mvit" "X3D_M 64 x3d"; do
  set -- $P
  timeout 600 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3.log 2>&1; echo "bench $3 rc=$?"; tail -1 gpurun_out/bench_$3.log | cut -c1-600
done
timeout 900 python bench.py --preset SLOWFAST_32x2_R101_50_50 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r101nl_b16.log 2>&1; echo "bench r101nl b16 rc=$?"; tail -1 gpurun_out/bench_r101nl_b16.log | cut -c1-600
rm -rf gpurun_out/pmc
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc -o fetch -- python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-profile > gpurun_out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc -o write -- python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-profile > gpurun_out/pmc_write.log 2>&1; echo "pmc write rc=$?"
python tools/pmc_traffic.py gpurun_out/pmc/fetch_counter_collection.csv gpurun_out/pmc/write_counter_collection.csv gpurun_out/pmc_traffic.json 2>&1 | tail -14
