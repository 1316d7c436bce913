#!/bin/bash
# round 2, GPU visit 4: rocprofv3 kernel trace of the default bench (in-graph per-kernel times) with igemm2 + materialisation.
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench4.log 2>&1; echo "bench rc=$? $(tail -1 gpurun_out/bench4.log | cut -c1-200)"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -o slowfast -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof4.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
F=$(find gpurun_out/prof4 -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" gpurun_out/r2_v4_slowfast_kernel_stats.md "round 2 visit 4: SlowFast-8x8-R50 bs32 default bench (igemm2 + materialised activations), rocprofv3 --kernel-trace --stats" 2>&1 | tail -2; head -40 gpurun_out/r2_v4_slowfast_kernel_stats.md
find gpurun_out/prof4 -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2_v4_kernel_trace.csv; ls -la gpurun_out/r2_v4_kernel_trace.csv
