#!/bin/bash
# round 3, visit 2: fused BatchNorm-backward reduction on hardware (kernel / block / model / step tests), A/B on the step,
# kernel trace (who issues the d2d copies), weight-gradient schedule sweep.
mkdir -p gpurun_out/v2
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SF_PARITY_REPORT=$R/gpurun_out/v2/parity_report.jsonl timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step.py "tests/test_model_gpu.py::test_blocks_strict" "tests/test_model_gpu.py::test_blocks_strict_x3d_nonlocal_mvit" "tests/test_model_gpu.py::test_well_conditioned_1e3_no_yardstick" "tests/test_model_gpu.py::test_full_size_batch2_against_oracle" -q -m gpu --tb=short -s -x > gpurun_out/v2/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/v2/pytest_gpu.log | tail -8 | cut -c1-400
B="python bench.py --no-secondary --no-cpu-baseline --no-kernel-profile --steps 10 --warmup 3"
for i in 1 2; do
  SF_BN_FUSE_REDUCE=0 timeout 200 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fuse=0', d['value'], d['ms_per_step'])"
  SF_BN_FUSE_REDUCE=1 timeout 200 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fuse=1', d['value'], d['ms_per_step'])"
done
SF_BN_FUSE_REDUCE=0 timeout 200 $B --preset X3D_M --batch 64 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('x3d fuse=0', d['value'], d['ms_per_step'])"
SF_BN_FUSE_REDUCE=1 timeout 200 $B --preset X3D_M --batch 64 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('x3d fuse=1', d['value'], d['ms_per_step'])"
timeout 300 python tools/wgrad_sweep.py --md gpurun_out/v2/wgrad_sweep.md > gpurun_out/v2/wgrad_sweep.log 2>&1; echo "sweep rc=$?"; tail -3 gpurun_out/v2/wgrad_sweep.log | cut -c1-300
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v2/prof -o p -- python $R/bench.py --no-secondary --no-cpu-baseline --no-kernel-profile --steps 3 --warmup 2 > $R/gpurun_out/v2/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $R
F=$(find gpurun_out/v2/prof -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" gpurun_out/v2/r3_v2_slowfast_kernel_stats.md "round 3 visit 2 (fused BN-backward reduce): slowfast default bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
T=$(find gpurun_out/v2/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_neighbors.py "$T" copyBuffer > gpurun_out/v2/r3_copybuffer_neighbors.txt 2>&1
head -30 gpurun_out/v2/r3_copybuffer_neighbors.txt | cut -c1-200
head -30 gpurun_out/v2/r3_v2_slowfast_kernel_stats.md | tail -22 | cut -c1-150
find gpurun_out/v2 -name "*.csv" -size +1M -delete
