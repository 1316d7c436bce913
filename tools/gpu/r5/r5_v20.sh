#!/bin/bash
# round 5 visit 20: Slow-pathway weight gradients enqueued on the Fast pathway's stream (two streams in all)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v20; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
ROUNDS=2 bash tools/gpu/ab.sh $D -- "base:X=1" "xpath:SF_WGRAD_XPATH=1" "xpath blocks256:SF_WGRAD_XPATH=1,SF_WGRAD2_BLOCKS=256" "xpath blocks384:SF_WGRAD_XPATH=1,SF_WGRAD2_BLOCKS=384"
ROUNDS=1 bash tools/gpu/ab.sh $D --no-graph -- "eager base:X=1" "eager xpath:SF_WGRAD_XPATH=1" "eager xpath blocks256:SF_WGRAD_XPATH=1,SF_WGRAD2_BLOCKS=256"
echo "exit 0"
