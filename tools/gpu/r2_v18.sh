#!/bin/bash
# round 2, visit 18: brick-tiled batched weight packing, fold threshold 2048; per-geometry table.
mkdir -p gpurun_out/v18
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_step.py tests/test_kernels_gpu.py -x -q -m gpu --tb=short > gpurun_out/v18/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/v18/pytest.log | cut -c1-300
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-secondary"
for P in "SLOWFAST_8x8_R50" "MVITv2_S_16x4" "X3D_M"; do
  timeout 300 $B --preset $P > gpurun_out/v18/bench_$P.json 2> gpurun_out/v18/bench_$P.err; echo "$P: $(python -c "import json;d=json.loads(open('gpurun_out/v18/bench_$P.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
done
SF_PACK_PLAN=0 timeout 300 $B > gpurun_out/v18/ab_noplan.json 2>/dev/null; echo "noplan: $(python -c "import json;d=json.loads(open('gpurun_out/v18/ab_noplan.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
timeout 300 $B > gpurun_out/v18/again.json 2>/dev/null; echo "again: $(python -c "import json;d=json.loads(open('gpurun_out/v18/again.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
timeout 600 python tools/microbench.py --iters 5 --md gpurun_out/v18/r2_v18_per_geometry.md --json gpurun_out/v18/microbench.json > gpurun_out/v18/microbench.txt 2>&1; echo "microbench rc=$?"; tail -22 gpurun_out/v18/r2_v18_per_geometry.md | cut -c1-150
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v18/prof -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary > $R/gpurun_out/v18/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $R
F=$(find gpurun_out/v18/prof -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" gpurun_out/v18/r2_v18_slowfast_kernel_stats.md "round 2 visit 18: slowfast bench (5 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
grep -E "prep_weights|finalize|part_fold|copyBuffer|rowtab" gpurun_out/v18/r2_v18_slowfast_kernel_stats.md | cut -c1-150
find gpurun_out/v18 -name "*.csv" -size +1M -delete
