#!/bin/bash
# round 5 visit 14: kernel stats of X3D-M (batch 64) and MViTv2-S at HEAD
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v14; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary"
for P in "X3D_M 64 x3d" "MVITv2_S_16x4 32 mvit"; do
  set -- $P
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/prof_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/$D/rocprof_$3.log 2>&1; echo "rocprof $3 rc=$?"
  F=$(find $R/$D/prof_$3 -name "*kernel_stats.csv" | head -1)
  python $R/tools/rocprof_summary.py "$F" $R/$D/r5_v14_${3}_kernel_stats.md "round 5 visit 14 (HEAD): $3 bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
  rm -rf $R/$D/prof_$3
done
head -45 $R/$D/r5_v14_x3d_kernel_stats.md | cut -c1-150
echo "exit 0"
