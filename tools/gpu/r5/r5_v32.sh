#!/bin/bash
# round 5 visit 32: BatchNorm streaming kernels two rows per iteration (loads first) against the previous commit, same box
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v32; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD

ROUNDS=2 bash tools/gpu/ab.sh $D -- "slowfast new:X=1" "slowfast prev:SFAMD_LIBRARY=$PWD/slowfast_amd/libsfamd_prev.so,SF_ALLOW_STALE_LIBRARY=1"
echo "exit 0"
