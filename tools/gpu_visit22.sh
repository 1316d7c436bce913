#!/bin/bash
# GPU visit 22: depthwise stencils with fp16 LDS weights (+ optional one-plane prefetch): parity + A/B.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --tb=short -k "dwconv or x3d or mvit or token" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu (subset) rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -8 | cut -c1-600
run() { local name=$1 preset=$2 batch=$3; shift 3
  env "$@" timeout 600 python bench.py --preset $preset --batch $batch --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$name.log 2>&1
  echo "$name rc=$? $(grep -h '^{' gpurun_out/ab_$name.log | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], {k:v["ms"] for k,v in j["kernels"].items() if "dw" in k})' 2>/dev/null)"; }
run x3d_pf1 X3D_M 64 SF_DW_PREFETCH=1
run x3d_pf0 X3D_M 64 SF_DW_PREFETCH=0
run mvit_pf1 MVITv2_S_16x4 32 SF_DW_PREFETCH=1
run mvit_pf0 MVITv2_S_16x4 32 SF_DW_PREFETCH=0
