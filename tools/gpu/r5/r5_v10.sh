#!/bin/bash
# round 5 visit 10: dispatch knobs re-swept UNDER pathway concurrency (their optima were found on one stream)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v10; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { L=$1; shift; env "$@" timeout 300 python bench.py $ARGS $B 2> $D/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a $D/ab.txt; }
ARGS=""
run "base" X=1
run "igemm3>=1" SF_IGEMM3=1
run "igemm3>=150" SF_IGEMM3=150
run "igemm3>=190" SF_IGEMM3=190
run "igemm3>=300" SF_IGEMM3=300
run "base" X=1
run "wgrad2 blocks 384" SF_WGRAD2_BLOCKS=384
run "wgrad2 blocks 448" SF_WGRAD2_BLOCKS=448
run "wgrad2 blocks 640" SF_WGRAD2_BLOCKS=640
run "wgrad2 blocks 768" SF_WGRAD2_BLOCKS=768
run "igemm2 bk 32" SF_IGEMM2_BK=32
run "igemm2 bk 64" SF_IGEMM2_BK=64
run "base" X=1
run "wgrad blocks 512" SF_WGRAD_BLOCKS=512
run "wgrad blocks 2048" SF_WGRAD_BLOCKS=2048
run "wgrad2t blocks 256" SF_WGRAD2T_BLOCKS=256
run "wgrad2t blocks 1024" SF_WGRAD2T_BLOCKS=1024
run "bn bwd blocks 512" SF_BN_BWD_BLOCKS=512
run "bn bwd blocks 2048" SF_BN_BWD_BLOCKS=2048
run "igemm occ4=0" SF_IGEMM_OCC4=0
run "base" X=1
echo "exit 0"
