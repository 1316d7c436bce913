#!/bin/bash
# round 5 visit 18: three branch streams (q | k | v) against two (q | k+v)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v18; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
ROUNDS=3 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit 2 branches:SF_BRANCHES3=0" "mvit 3 branches:SF_BRANCHES3=1"
echo "exit 0"
