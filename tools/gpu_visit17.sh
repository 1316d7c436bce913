#!/bin/bash
# GPU visit 17: GELU fused into the Mlp GEMM epilogues (sf_gemm_act, ABI v11): parity + A/B.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --tb=short -k "gelu or mvit or abi or tokens or gemm" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu (subset) rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -8 | cut -c1-600
for V in "1 fused" "0 unfused"; do
  set -- $V
  SF_GELU_FUSED=$1 timeout 600 python bench.py --preset MVITv2_S_16x4 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_mvit_gelu_$2.log 2>&1; echo "bench mvit gelu $2 rc=$?"; tail -1 gpurun_out/bench_mvit_gelu_$2.log | cut -c1-260
done
