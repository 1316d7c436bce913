#!/bin/bash
# round 5 visit 1: (a) new depthwise production-shape cases; (b) TIMING-ONLY ablation: what do the BatchNorm finalize launches
# cost inside the replayed step (upper bound for folding them into their producers)?  (c) s_setprio experiments in sf_igemm2
# (SF_IGEMM2_ABLATE bit 32 = static priority for waves 4-7, bit 64 = priority around every MFMA cluster)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v1; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 600 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py -k "dwconv" > $D/pytest_dw.log 2>&1; echo "pytest dw rc=$?"; tail -2 $D/pytest_dw.log | cut -c1-300
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { L=$1; shift; env "$@" timeout 300 python $TOOL $ARGS $B 2> $D/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt; grep ablate_calls $D/err.txt | tee -a $D/ab.txt; }
for R in 1 2; do
  ARGS=""; TOOL=bench.py
  run "slowfast base" X=1
  run "slowfast setprio-static" SF_IGEMM2_ABLATE=32
  run "slowfast setprio-cluster" SF_IGEMM2_ABLATE=64
  TOOL=tools/ablate_calls.py
  run "slowfast no-finalize" SF_SKIP_CALLS=sf_bn_finalize,sf_bn_bwd_finalize
  run "slowfast no-bn_bwd_finalize" SF_SKIP_CALLS=sf_bn_bwd_finalize
  ARGS="--preset MVITv2_S_16x4"; TOOL=bench.py
  run "mvit base" X=1
  run "mvit setprio-static" SF_IGEMM2_ABLATE=32
  run "mvit setprio-cluster" SF_IGEMM2_ABLATE=64
done
ARGS=""
for A in 0 32 64; do
SF_IGEMM2_ABLATE=$A timeout 300 python tools/microbench.py --no-bn --iters 8 --filter "slow" 2>&1 | grep -E "^s[2-5]" | cut -c1-170 | sed "s/^/ablate=$A /" | tee -a $D/microbench.txt
done
echo "exit 0"
