#!/bin/bash
# round 5 visit 6: MViTv2-S, Linear weight gradients on the side stream (SF_TOKEN_WGRAD_STREAM)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v6; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { L=$1; shift; env "$@" timeout 300 python bench.py $ARGS $B 2> $D/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a $D/ab.txt; }
for R in 1 2 3; do
  ARGS="--preset MVITv2_S_16x4"
  run "mvit wgrad-stream=0" SF_TOKEN_WGRAD_STREAM=0
  run "mvit wgrad-stream=1" SF_TOKEN_WGRAD_STREAM=1
done
ARGS="--preset X3D_M --batch 64"
run "x3d base" X=1
run "x3d wgrad-stream" SF_WGRAD_STREAM=1
tail -3 $D/err.txt | cut -c1-300
echo "exit 0"
