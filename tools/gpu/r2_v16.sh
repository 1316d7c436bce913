#!/bin/bash
# round 2, visit 16: launch-count cuts (wide finalize, cached row table, batched weight packing): tests, A/B, kernel stats.
mkdir -p gpurun_out/v16
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_step.py tests/test_kernels_gpu.py tests/test_data_parallel.py -x -q -m gpu --tb=short > gpurun_out/v16/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/v16/pytest.log | cut -c1-300
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/v16/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/v16/smoke.log | cut -c1-300
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-secondary"
for P in "SLOWFAST_8x8_R50" "MVITv2_S_16x4" "X3D_M"; do
  timeout 300 $B --preset $P > gpurun_out/v16/bench_$P.json 2> gpurun_out/v16/bench_$P.err; echo "$P new: $(python -c "import json;d=json.loads(open('gpurun_out/v16/bench_$P.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
done
SF_PACK_PLAN=0 timeout 300 $B > gpurun_out/v16/ab_noplan.json 2>/dev/null; echo "noplan: $(python -c "import json;d=json.loads(open('gpurun_out/v16/ab_noplan.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
SF_FOLD_ABOVE=256 timeout 300 $B > gpurun_out/v16/ab_fold.json 2>/dev/null; echo "fold(old): $(python -c "import json;d=json.loads(open('gpurun_out/v16/ab_fold.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
SF_WGRAD2_BLOCKS=512 timeout 300 $B > gpurun_out/v16/ab_w512.json 2>/dev/null; echo "wgrad2 512 blocks: $(python -c "import json;d=json.loads(open('gpurun_out/v16/ab_w512.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v16/prof -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary > $R/gpurun_out/v16/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $R
F=$(find gpurun_out/v16/prof -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" gpurun_out/v16/r2_v16_slowfast_kernel_stats.md "round 2 visit 16: slowfast bench (10 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
head -40 gpurun_out/v16/r2_v16_slowfast_kernel_stats.md | cut -c1-150
find gpurun_out/v16 -name "*.csv" -size +1M -delete
