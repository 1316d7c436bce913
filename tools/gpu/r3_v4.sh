#!/bin/bash
# round 3, visit 4: MViT dispatch thresholds + fc1 bias-gradient fusion A/B, L2 hit rates in the step (cache residency), per-step copyBuffer count
mkdir -p gpurun_out/v4
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_tokens_gpu.py "tests/test_model_gpu.py::test_blocks_strict_x3d_nonlocal_mvit" "tests/test_model_gpu.py::test_mvit_matches_reference" -q -m gpu --tb=short -x > gpurun_out/v4/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/v4/pytest_gpu.log | tail -4 | cut -c1-300
B="python bench.py --no-secondary --no-cpu-baseline --no-kernel-profile --steps 8 --warmup 3 --preset MVITv2_S_16x4"
run() { timeout 200 env "$@" $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit $*', d['value'], d['ms_per_step'])"; }
run SF_FUSE_COLSUM=0
run SF_FUSE_COLSUM=1
run SF_IGEMM2_MINK=384
run SF_IGEMM2_MINK=192
run SF_IGEMM2_MINK=96
run SF_WGRAD2_MINK=96
run SF_WGRAD2_BLOCKS=1024
run SF_FUSE_COLSUM=1
BX="python bench.py --no-secondary --no-cpu-baseline --no-kernel-profile --steps 8 --warmup 3 --preset X3D_M --batch 64"
timeout 200 $BX 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('x3d', d['value'], d['ms_per_step'])"
cd /tmp
BS="python $R/bench.py --no-secondary --no-cpu-baseline --no-kernel-profile --steps 3 --warmup 2"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/v4/pmc_tcc -o p -- $BS > $R/gpurun_out/v4/pmc_tcc.log 2>&1; echo "pmc tcc rc=$?"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/v4/trace -o p -- $BS > $R/gpurun_out/v4/trace.log 2>&1; echo "trace rc=$?"
cd $R
FM=$(find gpurun_out/v4/pmc_tcc -name "*counter_collection.csv" | head -1)
python tools/pmc_metric.py gpurun_out/v4/r3_v4_pmc_tcc_slowfast.md "round 3 visit 4: L2 (TCC) hits / misses per kernel inside the SlowFast step" "$FM" > /dev/null 2>&1
head -30 gpurun_out/v4/r3_v4_pmc_tcc_slowfast.md | cut -c1-170
T=$(find gpurun_out/v4/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_neighbors.py "$T" copyBuffer > gpurun_out/v4/r3_v4_copybuffer_per_step.txt 2>&1
grep "step between" gpurun_out/v4/r3_v4_copybuffer_per_step.txt | cut -c1-200
find gpurun_out/v4 -name "*.csv" -size +1M -delete
