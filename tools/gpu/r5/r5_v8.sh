#!/bin/bash
# round 5 visit 8: Fast pathway running ahead (no fork wait when its input is its own stream's product)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v8; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_step.py -k "pathway or segmented or poisoned" > $D/pytest_step.log 2>&1; echo "pytest step rc=$?"; tail -3 $D/pytest_step.log | cut -c1-300
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { L=$1; shift; env "$@" timeout 300 python bench.py $ARGS $B 2> $D/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a $D/ab.txt; }
for R in 1 2 3; do
  ARGS=""
  run "slowfast ahead=1" SF_PATHWAY_RUN_AHEAD=1
  run "slowfast ahead=0" SF_PATHWAY_RUN_AHEAD=0
done
ARGS="--preset SLOWFAST_32x2_R101_50_50 --batch 8"
run "r101nl ahead=1" SF_PATHWAY_RUN_AHEAD=1
run "r101nl ahead=0" SF_PATHWAY_RUN_AHEAD=0
echo "exit 0"
