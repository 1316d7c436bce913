#!/bin/bash
# round 5 visit 29: visit 28 without the pipelined GEMM store loops (attention changes only)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v29; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit:X=1"
ROUNDS=2 bash tools/gpu/ab.sh $D -- "slowfast:X=1"
echo "exit 0"
