#!/bin/bash
# round 5 visit 12: kernel trace of the two-stream SlowFast step: kernel stats + queue timeline
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v12; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/prof_slowfast -o p -- $BENCH > $R/$D/rocprof_slowfast.log 2>&1; echo "rocprof rc=$?"
cd $R
F=$(find $D/prof_slowfast -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" $D/r5_v12_slowfast_kernel_stats.md "round 5 visit 12 (two pathway streams): slowfast default bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
T=$(find $D/prof_slowfast -name "*kernel_trace.csv" | head -1)
head -2 "$T" | cut -c1-400
python tools/stream_timeline.py "$T" $D/r5_v12_slowfast_timeline.md --last-frac 0.45 | head -60
rm -rf $D/prof_slowfast
echo "exit 0"
