#!/bin/bash
# round 4 visit 25: rocprofv3 kernel table of the MViTv2-S bench command at HEAD (the last GPU seconds of the round)
D=gpurun_out/v25; mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/prof_mvit -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary --preset MVITv2_S_16x4 --batch 32 > $R/$D/rocprof_mvit.log 2>&1; echo "rocprof rc=$?"
cd $R
F=$(find $D/prof_mvit -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" $D/r4_v25_mvit_kernel_stats.md "round 4 visit 25 (HEAD): MViTv2-S bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
head -12 $D/r4_v25_mvit_kernel_stats.md | cut -c1-150
