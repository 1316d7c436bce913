#!/bin/bash
# round 3 visit 10: three-stage ring of the pointwise (direct-to-LDS) path of sf_igemm_kernel done properly (asm copies, raw
# barrier, counted waits): SF_IGEMM_GL3=1 vs 0 -- correctness, MViT Linear shapes, SlowFast pointwise layers, whole steps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/v10; export TMPDIR=/tmp
SF_IGEMM_GL3=1 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_tokens_gpu.py -x -q -m gpu > gpurun_out/v10/pytest_gpu.log 2>&1
echo "pytest gpu (gl3=1) rc=$?"; tail -2 gpurun_out/v10/pytest_gpu.log
OUT=gpurun_out/v10/gl3_ab.txt
: > $OUT
for V in 0 1; do
  echo "== gemm_bench SF_IGEMM_GL3=$V" | tee -a $OUT
  SF_IGEMM_GL3=$V timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "^\| (s[1-4]|\*\*)" | tee -a $OUT
  echo "== microbench SF_IGEMM_GL3=$V" | tee -a $OUT
  SF_IGEMM_GL3=$V timeout 300 python tools/microbench.py --no-bn --iters 8 --filter "slow a|slow c|fast c|fuse" 2>&1 | grep -E "^(s[2-5]|fuse)" | cut -c1-125 | tee -a $OUT
done
B="python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
for R in 1 2; do
  for V in 0 1; do
    SF_IGEMM_GL3=$V timeout 200 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('slowfast gl3=$V', d['value'], d['ms_per_step'])" | tee -a $OUT
    SF_IGEMM_GL3=$V timeout 200 $B --preset MVITv2_S_16x4 --steps 10 --warmup 3 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit gl3=$V', d['value'], d['ms_per_step'])" | tee -a $OUT
  done
done
echo "exit 0"
