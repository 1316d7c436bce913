#!/bin/bash
# round 5 visit 9: stream priority experiment (Slow pathway on a high-priority stream)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v9; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { L=$1; shift; env "$@" timeout 300 python bench.py $ARGS $B 2> $D/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a $D/ab.txt; }
for R in 1 2; do
  ARGS=""
  run "slowfast prio=0" SF_PATHWAY_PRIO=0
  run "slowfast prio=1" SF_PATHWAY_PRIO=1
  ARGS="--no-graph"
  run "slowfast eager prio=0" SF_PATHWAY_PRIO=0
  run "slowfast eager prio=1" SF_PATHWAY_PRIO=1
done
tail -3 $D/err.txt | cut -c1-200
echo "exit 0"
