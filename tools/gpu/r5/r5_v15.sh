#!/bin/bash
# round 5 visit 15: lateral concat written in place (SF_CAT_IN_PLACE) A/B + model parity
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v15; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_model_gpu.py -k "slowfast or r101" > $D/pytest_model.log 2>&1; echo "pytest model rc=$?"; tail -3 $D/pytest_model.log | cut -c1-300
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { L=$1; shift; env "$@" timeout 300 python bench.py $ARGS $B 2> $D/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a $D/ab.txt; }
for R in 1 2 3; do
  ARGS=""
  run "slowfast cat-in-place=1" SF_CAT_IN_PLACE=1
  run "slowfast cat-in-place=0" SF_CAT_IN_PLACE=0
done
echo "exit 0"
