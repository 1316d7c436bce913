#!/bin/bash
# GPU visit 19: incremental (LEAN) GEMM loader A/B + parity of the conv / token kernels with it.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --tb=short -k "conv or kernels or blocks or gemm or stem" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu (conv subset) rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -8 | cut -c1-600
run() { local name=$1 preset=$2 batch=$3; shift 3
  env "$@" timeout 600 python bench.py --preset $preset --batch $batch --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/ab_$name.log 2>&1
  echo "$name rc=$? $(grep -h '^{' gpurun_out/ab_$name.log | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])' 2>/dev/null)"; }
run sf_lean1 SLOWFAST_8x8_R50 32 SF_IGEMM_LEAN=1
run sf_lean0 SLOWFAST_8x8_R50 32 SF_IGEMM_LEAN=0
run x3d_lean1 X3D_M 64 SF_IGEMM_LEAN=1
run x3d_lean0 X3D_M 64 SF_IGEMM_LEAN=0
run mvit_lean1 MVITv2_S_16x4 32 SF_IGEMM_LEAN=1
run mvit_lean0 MVITv2_S_16x4 32 SF_IGEMM_LEAN=0
