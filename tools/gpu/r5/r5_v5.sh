#!/bin/bash
# round 5 visit 5: pathway streams (engine.run_pathways), productized: bitwise tests + A/B with loss equality at full size
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v5; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_step.py > $D/pytest_step.log 2>&1; echo "pytest step rc=$?"; tail -4 $D/pytest_step.log | cut -c1-300
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_model_gpu.py -k "slowfast or r101 or c2d" > $D/pytest_model.log 2>&1; echo "pytest model rc=$?"; tail -4 $D/pytest_model.log | cut -c1-300
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { L=$1; shift; env "$@" timeout 300 python bench.py $ARGS $B 2> $D/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a $D/ab.txt; }
for R in 1 2 3; do
  ARGS=""
  run "slowfast streams=1" SF_PATHWAY_STREAMS=1
  run "slowfast streams=0" SF_PATHWAY_STREAMS=0
done
ARGS="--preset SLOWFAST_32x2_R101_50_50 --batch 8"
run "r101nl streams=1" SF_PATHWAY_STREAMS=1
run "r101nl streams=0" SF_PATHWAY_STREAMS=0
run "r101nl streams=1" SF_PATHWAY_STREAMS=1
ARGS="--no-graph"
run "slowfast eager streams=1" SF_PATHWAY_STREAMS=1
run "slowfast eager streams=0" SF_PATHWAY_STREAMS=0
echo "exit 0"
