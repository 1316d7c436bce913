#!/bin/bash
# round 2, visit 19: BN streaming kernels with two rows in flight per thread, 2048 reduce blocks; division-free pack kernel.
mkdir -p gpurun_out/v19
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_step.py tests/test_kernels_gpu.py -x -q -m gpu --tb=short > gpurun_out/v19/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/v19/pytest.log | cut -c1-300
timeout 600 python tools/microbench.py --iters 5 --md gpurun_out/v19/r2_v19_per_geometry.md > gpurun_out/v19/microbench.txt 2>&1; echo "microbench rc=$?"; grep "weighted totals" gpurun_out/v19/r2_v19_per_geometry.md
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-secondary"
for P in "SLOWFAST_8x8_R50" "MVITv2_S_16x4"; do
  timeout 300 $B --preset $P > gpurun_out/v19/bench_$P.json 2> gpurun_out/v19/bench_$P.err; echo "$P: $(python -c "import json;d=json.loads(open('gpurun_out/v19/bench_$P.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
done
timeout 300 $B --preset X3D_M --batch 64 > gpurun_out/v19/bench_X3D_M.json 2>/dev/null; echo "X3D b64: $(python -c "import json;d=json.loads(open('gpurun_out/v19/bench_X3D_M.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
SF_BN_BWD_BLOCKS=1024 timeout 300 $B > gpurun_out/v19/ab_blocks1024.json 2>/dev/null; echo "reduce 1024 blocks: $(python -c "import json;d=json.loads(open('gpurun_out/v19/ab_blocks1024.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
SF_BN_BWD_BLOCKS=4096 SF_FOLD_ABOVE=4096 timeout 300 $B > gpurun_out/v19/ab_blocks4096.json 2>/dev/null; echo "reduce 4096 blocks: $(python -c "import json;d=json.loads(open('gpurun_out/v19/ab_blocks4096.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v19/prof -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary > $R/gpurun_out/v19/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $R
F=$(find gpurun_out/v19/prof -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" gpurun_out/v19/r2_v19_slowfast_kernel_stats.md "round 2 visit 19: slowfast bench (5 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
grep -E "sf_bn_|prep_weights|part_fold" gpurun_out/v19/r2_v19_slowfast_kernel_stats.md | cut -c1-150
find gpurun_out/v19 -name "*.csv" -size +1M -delete
