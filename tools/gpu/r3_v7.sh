#!/bin/bash
# round 3 visit 7: staggered copy issue in sf_igemm2_kernel (SF_IGEMM2_STAGGER=1) -- correctness on the GPU, per layer, whole step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/v7; export TMPDIR=/tmp
SF_IGEMM2_STAGGER=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/v7/pytest_gpu.log 2>&1
echo "pytest gpu (stagger=1) rc=$?"; tail -2 gpurun_out/v7/pytest_gpu.log
OUT=gpurun_out/v7/stagger_ab.txt
: > $OUT
for V in 0 1; do
  echo "== microbench SF_IGEMM2_STAGGER=$V" | tee -a $OUT
  SF_IGEMM2_STAGGER=$V timeout 300 python tools/microbench.py --no-bn --iters 8 --filter "slow b|slow a|slow c" 2>&1 | grep -E "^s[2-5]" | cut -c1-125 | tee -a $OUT
done
B="python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
for R in 1 2; do
  for V in 0 1; do
    SF_IGEMM2_STAGGER=$V timeout 200 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('stagger=$V', d['value'], d['ms_per_step'])" | tee -a $OUT
  done
done
echo "exit 0"
