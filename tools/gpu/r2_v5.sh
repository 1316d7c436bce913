#!/bin/bash
# round 2, GPU visit 5: strided dgrad as residue-class sub-convolutions + 1-bit ReLU masks in BatchNorm backward: parity, A/B.
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels_gpu.py -q --tb=short -k "igemm2 or conv_dgrad or bn_chain" > gpurun_out/pytest5.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest5.log | tail -8 | cut -c1-300
timeout 400 python -m pytest tests/test_model_gpu.py -q --tb=short -k "blocks_strict or model_matches" > gpurun_out/pytest5b.log 2>&1; echo "pytest models rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest5b.log | tail -8 | cut -c1-300
for V in "default SF_X=0" "nostrided SF_IGEMM2_STRIDED=0"; do
  set -- $V
  env $2 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench5_$1.log 2>&1; echo "bench $1 rc=$? $(tail -1 gpurun_out/bench5_$1.log | cut -c1-200)"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof5 -o slowfast -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof5.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
F=$(find gpurun_out/prof5 -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" gpurun_out/r2_v5_slowfast_kernel_stats.md "round 2 visit 5: SlowFast-8x8-R50 bs32 default bench, rocprofv3 --kernel-trace --stats" 2>&1 | tail -1; head -34 gpurun_out/r2_v5_slowfast_kernel_stats.md | tail -26
