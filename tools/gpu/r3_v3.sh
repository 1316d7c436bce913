#!/bin/bash
# round 3, visit 3: two-rows-in-flight epilogue + split rounding: correctness (kernels, blocks), A/B on the step, wgrad sweep, copyBuffer context.
mkdir -p gpurun_out/v3
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_tokens_gpu.py "tests/test_model_gpu.py::test_blocks_strict" "tests/test_model_gpu.py::test_blocks_strict_x3d_nonlocal_mvit" "tests/test_model_gpu.py::test_well_conditioned_1e3_no_yardstick" -q -m gpu --tb=short -x > gpurun_out/v3/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/v3/pytest_gpu.log | tail -6 | cut -c1-400
B="python bench.py --no-secondary --no-cpu-baseline --no-kernel-profile --steps 10 --warmup 3"
for i in 1 2; do
  SF_BN_FUSE_REDUCE=0 timeout 200 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fuse=0', d['value'], d['ms_per_step'])"
  SF_BN_FUSE_REDUCE=1 timeout 200 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fuse=1', d['value'], d['ms_per_step'])"
done
SF_WGRAD2_BLOCKS=1024 timeout 200 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fuse=1 wgrad2 blocks 1024', d['value'], d['ms_per_step'])"
SF_WGRAD2_BLOCKS=384 timeout 200 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fuse=1 wgrad2 blocks 384', d['value'], d['ms_per_step'])"
timeout 200 $B --preset MVITv2_S_16x4 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit', d['value'], d['ms_per_step'])"
timeout 300 python tools/wgrad_sweep.py --md gpurun_out/v3/wgrad_sweep.md > gpurun_out/v3/wgrad_sweep.log 2>&1; echo "sweep rc=$?"; tail -2 gpurun_out/v3/wgrad_sweep.log | cut -c1-300
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v3/prof -o p -- python $R/bench.py --no-secondary --no-cpu-baseline --no-kernel-profile --steps 3 --warmup 2 > $R/gpurun_out/v3/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $R
F=$(find gpurun_out/v3/prof -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" gpurun_out/v3/r3_v3_slowfast_kernel_stats.md "round 3 visit 3 (fused BN-backward reduce, two-row epilogue, wgrad split rounding): slowfast default bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
T=$(find gpurun_out/v3/prof -name "*kernel_trace.csv" | head -1)
head -1 "$T" > gpurun_out/v3/trace_header.txt
python tools/trace_neighbors.py "$T" copyBuffer > gpurun_out/v3/r3_copybuffer_neighbors.txt 2>&1
head -24 gpurun_out/v3/r3_copybuffer_neighbors.txt | cut -c1-200
head -26 gpurun_out/v3/r3_v3_slowfast_kernel_stats.md | tail -18 | cut -c1-150
find gpurun_out/v3 -name "*.csv" -size +1M -delete
