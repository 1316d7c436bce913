#!/bin/bash
# round 5 visit 22: MViT Linear weight gradients on the branch stream (two streams in all) / on their own stream, with branches on
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v22; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "base:X=1" "wgrad own stream:SF_TOKEN_WGRAD_STREAM=1" "wgrad on branch stream:SF_TOKEN_WGRAD_STREAM=2"
echo "exit 0"
