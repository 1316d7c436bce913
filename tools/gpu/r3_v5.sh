#!/bin/bash
# round 3, visit 5: the STRIP variant of the second-generation implicit GEMM (one staged strip per channel chunk serves all taps)
mkdir -p gpurun_out/v5
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py "tests/test_model_gpu.py::test_blocks_strict" "tests/test_model_gpu.py::test_well_conditioned_1e3_no_yardstick" -q -m gpu --tb=short -x > gpurun_out/v5/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/v5/pytest_gpu.log | tail -4 | cut -c1-300
for V in 0 1 2; do
  echo "== microbench SF_IGEMM2_STRIP=$V"
  SF_IGEMM2_STRIP=$V timeout 200 python tools/microbench.py --no-bn --iters 6 --filter "slow b" 2>&1 | grep -E "slow b" | cut -c1-150
  SF_IGEMM2_STRIP=$V timeout 200 python tools/microbench.py --no-bn --iters 6 --filter "s5.slow a" 2>&1 | grep -E "slow a" | cut -c1-150
done
B="python bench.py --no-secondary --no-cpu-baseline --no-kernel-profile --steps 10 --warmup 3"
for i in 1 2; do
  for V in 0 1 2; do
    SF_IGEMM2_STRIP=$V timeout 200 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('strip=$V', d['value'], d['ms_per_step'])"
  done
done
