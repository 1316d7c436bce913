#!/bin/bash
# round 2, GPU visit 6: second-generation weight gradient (sf_wgrad2.h): parity, per-layer and end-to-end A/B.
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels_gpu.py -q --tb=short -k "igemm2 or conv_wgrad or wgrad_many" > gpurun_out/pytest6.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest6.log | tail -8 | cut -c1-300
run_mb() { local name=$1; shift; env "$@" timeout 120 python tools/microbench.py --json gpurun_out/mb6_$name.json --no-bn > gpurun_out/mb6_$name.txt 2>&1; echo "mb $name rc=$? $(tail -1 gpurun_out/mb6_$name.txt)"; }
run_mb w2 SF_X=0
run_mb w1 SF_WGRAD2=0
run_mb w2_b1024 SF_WGRAD2_BLOCKS=1024
run_mb w2_mink64 SF_WGRAD2_MINK=64
for V in "w2 SF_X=0" "w1 SF_WGRAD2=0" "w2_b1024 SF_WGRAD2_BLOCKS=1024" "w2_mink64 SF_WGRAD2_MINK=64"; do
  set -- $V
  env $2 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench6_$1.log 2>&1; echo "bench $1 rc=$? $(tail -1 gpurun_out/bench6_$1.log | cut -c1-200)"
done
