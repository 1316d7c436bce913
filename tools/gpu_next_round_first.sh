#!/bin/bash
# First GPU visit of the NEXT round (the round-1 GPU budget ended at visit 28): full gpu suite, smoke, default bench with
# cpu_baseline, the other presets, eval-path throughput, rocprofv3 kernel stats of the default bench command.  ~9 GPU-minutes.
#   gpurun --timeout 900 -- 'bash tools/gpu_next_round_first.sh'
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -8 | cut -c1-600
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 400 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "bench default rc=$?"; tail -1 gpurun_out/bench_default.log | cut -c1-700
for P in "MVITv2_S_16x4 32 mvit" "X3D_M 64 x3d" "C2D_8x8_R50 32 c2d" "SLOWFAST_32x2_R101_50_50 16 ava" "MVIT_B_16x4_CONV 32 mvit_v1"; do
  set -- $P
  timeout 200 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3.log 2>&1; echo "bench $3 rc=$?"; tail -1 gpurun_out/bench_$3.log | cut -c1-330
done
timeout 100 python tools/bench_eval.py --steps 5 > gpurun_out/bench_eval_slowfast.log 2>&1; tail -1 gpurun_out/bench_eval_slowfast.log | cut -c1-300
timeout 100 python tools/bench_eval.py --preset X3D_M --batch 64 --steps 5 > gpurun_out/bench_eval_x3d.log 2>&1; tail -1 gpurun_out/bench_eval_x3d.log | cut -c1-300
# A/B of the opt-in version-2 depthwise stencils (profiles/r1_isa_dwconv_v2.md)
for P in "X3D_M 64 x3d" "MVITv2_S_16x4 32 mvit"; do
  set -- $P
  SF_DW_FWD_V2=1 SF_DW_DGRAD_V2=1 SF_DW_WGRAD_V2=1 timeout 200 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3_dwv2.log 2>&1
  echo "bench $3 dw-v2 rc=$?"; tail -1 gpurun_out/bench_$3_dwv2.log | cut -c1-330
done
# A/B of the opt-in three-stage direct-to-LDS implicit GEMM (SF_IGEMM_GL3=1, DESIGN.md 7: memory-level-parallelism hypothesis)
for P in "SLOWFAST_8x8_R50 32 slowfast" "MVITv2_S_16x4 32 mvit"; do
  set -- $P
  SF_IGEMM_GL3=1 timeout 200 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3_gl3.log 2>&1
  echo "bench $3 gl3 rc=$?"; tail -1 gpurun_out/bench_$3_gl3.log | cut -c1-330
done
# option families added after the round-1 GPU budget ended (hostsim-green): first timing of the larger MViT presets
for P in "MVIT_B_16x4_CONV 32 mvit_v1" "REV_MVIT_B_16x4_CONV 32 rev_mvit"; do
  set -- $P
  timeout 200 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$3.log 2>&1
  echo "bench $3 rc=$?"; tail -1 gpurun_out/bench_$3.log | cut -c1-330
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o slowfast -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof_slowfast.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
F=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" gpurun_out/slowfast_kernel_stats.md "SlowFast-8x8-R50 bs32 default bench, rocprofv3 --kernel-trace --stats" 2>&1 | tail -2; head -20 gpurun_out/slowfast_kernel_stats.md
