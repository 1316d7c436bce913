#!/bin/bash
# round 4 visit 11: 256 x 256 tiles on a 16-wave workgroup (SF_IGEMM2_FAT=<min tiles>), in-step A/B on both models
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v11; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
SF_IGEMM2_FAT=1 timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_kernels_gpu.py tests/test_model_gpu.py -k "igemm2 or conv_fwd or conv_dgrad or slowfast_wc" > $D/pytest_fat.log 2>&1; echo "pytest fat rc=$?"; tail -2 $D/pytest_fat.log | cut -c1-300
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { L=$1; shift; env "$@" timeout 300 python bench.py $ARGS $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt; }
for R in 1 2; do
  ARGS=""
  run "slowfast base" X=1
  run "slowfast fat>=1" SF_IGEMM2_FAT=1
  run "slowfast fat>=200" SF_IGEMM2_FAT=200
  run "slowfast fat>=500" SF_IGEMM2_FAT=500
  ARGS="--preset MVITv2_S_16x4"
  run "mvit base" X=1
  run "mvit fat>=1" SF_IGEMM2_FAT=1
  run "mvit fat>=1 mink=384" SF_IGEMM2_FAT=1 SF_IGEMM2_MINK=384
done
ARGS=""
SF_IGEMM2_FAT=1 timeout 300 python tools/microbench.py --no-bn --iters 8 --filter "s4.slow" 2>&1 | grep -E "^s[2-5]" | cut -c1-170 | sed "s/^/fat=1 /" | tee $D/microbench.txt
timeout 300 python tools/microbench.py --no-bn --iters 8 --filter "s4.slow" 2>&1 | grep -E "^s[2-5]" | cut -c1-170 | sed "s/^/fat=0 /" | tee -a $D/microbench.txt
echo "exit 0"
