#!/bin/bash
# round 5 visit 2: in-launch BatchNorm statistics finalize (forward half): parity + in-step A/B
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v2; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_kernels_gpu.py -k "fold or conv_fwd or bn_chain" > $D/pytest_k.log 2>&1; echo "pytest kernels rc=$?"; tail -3 $D/pytest_k.log | cut -c1-300
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_model_gpu.py -k "resblock or slowfast_wc or stem or fuse" > $D/pytest_m.log 2>&1; echo "pytest model rc=$?"; tail -3 $D/pytest_m.log | cut -c1-300
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { L=$1; shift; env "$@" timeout 300 python bench.py $ARGS $B 2> $D/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a $D/ab.txt; }
for R in 1 2 3; do
  ARGS=""
  run "slowfast fold=1" SF_BN_FOLD=1
  run "slowfast fold=0" SF_BN_FOLD=0
done
ARGS="--preset SLOWFAST_32x2_R101_50_50 --batch 8"
run "r101nl fold=1" SF_BN_FOLD=1
run "r101nl fold=0" SF_BN_FOLD=0
ARGS="--preset C2D_8x8_R50"
run "c2d fold=1" SF_BN_FOLD=1
run "c2d fold=0" SF_BN_FOLD=0
echo "exit 0"
