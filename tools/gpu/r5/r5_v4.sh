#!/bin/bash
# round 5 visit 4: concurrency experiments (timing only): Fast pathway of every stage on a side stream, weight gradients on a side stream
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v4; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { L=$1; shift; env "$@" timeout 300 python bench.py $ARGS $B 2> $D/err_$2.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a $D/ab.txt; }
for R in 1 2; do
  ARGS=""
  run "slowfast base" X=1
  run "slowfast pathway-streams" SF_PATHWAY_STREAMS=1
  run "slowfast wgrad-stream" SF_WGRAD_STREAM=1
  run "slowfast both" SF_PATHWAY_STREAMS=1 SF_WGRAD_STREAM=1
  run "slowfast pathway-streams nograph" SF_PATHWAY_STREAMS=1 EXTRA=1
done
ARGS="--no-graph"
run "slowfast eager base" X=1
run "slowfast eager pathway-streams" SF_PATHWAY_STREAMS=1
tail -5 $D/err_*.txt | cut -c1-300
echo "exit 0"
