#!/bin/bash
# round 2, GPU visit 1: reference-derived (autocast) parity yardstick, the GPU tests the round-1 driver run never reached,
# per-layer microbench baseline, A/B of the opt-in round-1 kernels (three-stage direct-to-LDS GEMM, version-2 stencils).
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 420 python tools/autocast_yardstick.py --out gpurun_out/autocast_yardstick.json > gpurun_out/autocast.log 2>&1; echo "autocast rc=$?"; tail -3 gpurun_out/autocast.log | cut -c1-400
timeout 300 python -m pytest tests/test_zy_new_families_gpu.py tests/test_zz_optin_gpu.py -q --tb=short > gpurun_out/pytest_new.log 2>&1; echo "pytest new rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_new.log | tail -8 | cut -c1-300
timeout 200 python tools/microbench.py --json gpurun_out/microbench_base.json > gpurun_out/microbench_base.txt 2>&1; echo "microbench rc=$?"; tail -1 gpurun_out/microbench_base.txt
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_base.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_base.log | cut -c1-300
SF_IGEMM_GL3=1 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_gl3.log 2>&1; echo "bench gl3 rc=$?"; tail -1 gpurun_out/bench_gl3.log | cut -c1-200
for P in "X3D_M 64 x3d" "MVITv2_S_16x4 32 mvit"; do
  set -- $P
  timeout 150 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_$3.log 2>&1; echo "bench $3 rc=$?"; tail -1 gpurun_out/bench_$3.log | cut -c1-200
  SF_DW_FWD_V2=1 SF_DW_DGRAD_V2=1 SF_DW_WGRAD_V2=1 timeout 150 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_$3_dwv2.log 2>&1; echo "bench $3 dwv2 rc=$?"; tail -1 gpurun_out/bench_$3_dwv2.log | cut -c1-200
done
