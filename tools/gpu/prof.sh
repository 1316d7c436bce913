#!/bin/bash
# rocprofv3 kernel stats (+ queue timeline) of the bench command of one preset, run on the GPU box through gpurun:
#   tools/gpu/prof.sh OUTDIR PRESET BATCH [env K=V ...]       e.g.  tools/gpu/prof.sh gpurun_out/p1 MVITv2_S_16x4 32
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD; D=$R/$1; P=$2; B=$3; shift 3; mkdir -p "$D"
cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary --preset $P --batch $B > $D/rocprof.log 2>&1
echo "rocprof rc=$?"; tail -1 $D/rocprof.log | cut -c1-300
cd $R
F=$(find $D/prof -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" $D/kernel_stats.md "$P batch $B $*: bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
T=$(find $D/prof -name "*kernel_trace.csv" | head -1)
python tools/stream_timeline.py "$T" $D/timeline.md > /dev/null 2>&1
rm -rf $D/prof
head -45 $D/kernel_stats.md | cut -c1-150
head -12 $D/timeline.md | cut -c1-160
