#!/bin/bash
# round 2, GPU visit 14: state check after the register fix (SlowFast bench) + rocprofv3 kernel stats of MViTv2-S and X3D-M.
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/bench14_slowfast.log 2>&1; echo "bench slowfast rc=$? $(tail -1 gpurun_out/bench14_slowfast.log | cut -c1-200)"
for P in "MVITv2_S_16x4 32 mvit" "X3D_M 64 x3d"; do
  set -- $P
  timeout 150 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench14_$3.log 2>&1; echo "bench $3 rc=$? $(tail -1 gpurun_out/bench14_$3.log | cut -c1-200)"
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof14_$3 -o p -- python $GRAFT_REPO_ROOT/bench.py --preset $1 --batch $2 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof14_$3.log 2>&1; echo "rocprof $3 rc=$?"
  cd $GRAFT_REPO_ROOT
  F=$(find gpurun_out/prof14_$3 -name "*kernel_stats.csv" | head -1)
  python tools/rocprof_summary.py "$F" gpurun_out/r2_v14_$3_kernel_stats.md "round 2 visit 14: $1 bs$2 bench, rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
  head -30 gpurun_out/r2_v14_$3_kernel_stats.md | tail -23 | cut -c1-200
  find gpurun_out/prof14_$3 -name "*.csv" -size +1M -delete
done
