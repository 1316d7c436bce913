#!/bin/bash
# round 5 visit 33: MViTv2-S kernel table + queue timeline after the attention changes
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v33; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/prof_mvit -o p -- $BENCH --preset MVITv2_S_16x4 --batch 32 > $R/$D/rocprof_mvit.log 2>&1; echo "rocprof rc=$?"
cd $R
F=$(find $D/prof_mvit -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" $D/r5_v33_mvit_kernel_stats.md "round 5 (visit 33): mvit bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
T=$(find $D/prof_mvit -name "*kernel_trace.csv" | head -1)
python tools/stream_timeline.py "$T" $D/r5_v33_mvit_timeline.md > /dev/null 2>&1
head -45 $D/r5_v33_mvit_kernel_stats.md | cut -c1-150
head -40 $D/r5_v33_mvit_timeline.md | cut -c1-150
rm -rf $D/prof_mvit
echo "exit 0"
