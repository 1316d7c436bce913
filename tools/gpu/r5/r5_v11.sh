#!/bin/bash
# round 5 visit 11: MViTv2-S: Linear weight gradients on a side stream with HALF-occupancy grids (one workgroup per CU)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v11; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { L=$1; shift; env "$@" timeout 300 python bench.py $ARGS $B 2> $D/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a $D/ab.txt; }
ARGS="--preset MVITv2_S_16x4"
run "mvit base" X=1
run "mvit side, blocks 512" SF_TOKEN_WGRAD_STREAM=1
run "mvit side, blocks 256" SF_TOKEN_WGRAD_STREAM=1 SF_WGRAD2_BLOCKS=256 SF_WGRAD_BLOCKS=256
run "mvit side, blocks 128" SF_TOKEN_WGRAD_STREAM=1 SF_WGRAD2_BLOCKS=128 SF_WGRAD_BLOCKS=128
run "mvit main, blocks 256" SF_WGRAD2_BLOCKS=256 SF_WGRAD_BLOCKS=256
run "mvit base" X=1
ARGS="--preset MVITv2_S_16x4 --no-graph"
run "mvit eager base" X=1
run "mvit eager side, blocks 256" SF_TOKEN_WGRAD_STREAM=1 SF_WGRAD2_BLOCKS=256 SF_WGRAD_BLOCKS=256
echo "exit 0"
