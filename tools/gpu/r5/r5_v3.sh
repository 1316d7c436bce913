#!/bin/bash
# round 5 visit 3: sf_igemm3 (256 x 256 x 64, eight-phase ping-pong): parity + per-layer A/B + in-step A/B
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v3; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_igemm3_gpu.py > $D/pytest.log 2>&1; echo "pytest igemm3 rc=$?"; tail -5 $D/pytest.log | cut -c1-300
for A in 0 1; do
SF_IGEMM3=$A timeout 300 python tools/microbench.py --no-bn --iters 8 --filter "s3.slow c|s3.slow sc|s4.slow|s5.slow" 2>&1 | grep -E "^s[2-5]" | cut -c1-170 | sed "s/^/igemm3=$A /" | tee -a $D/microbench.txt
done
for A in 0 1; do
SF_IGEMM3=$A SF_IGEMM2_MINK=${MINK:-512} timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "^\| s[234]" | sed "s/^/igemm3=$A /" | tee -a $D/gemm_bench.txt
done
SF_IGEMM3=1 SF_IGEMM2_MINK=384 timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "^\| s[34]" | sed "s/^/igemm3=1,mink=384 /" | tee -a $D/gemm_bench.txt
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { L=$1; shift; env "$@" timeout 300 python bench.py $ARGS $B 2> $D/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a $D/ab.txt; }
for R in 1 2; do
  ARGS=""
  run "slowfast base" SF_IGEMM3=0
  run "slowfast igemm3>=1" SF_IGEMM3=1
  run "slowfast igemm3>=300" SF_IGEMM3=300
  ARGS="--preset MVITv2_S_16x4"
  run "mvit base" SF_IGEMM3=0
  run "mvit igemm3>=1" SF_IGEMM3=1
  run "mvit igemm3>=1 mink=384" SF_IGEMM3=1 SF_IGEMM2_MINK=384
done
echo "exit 0"
