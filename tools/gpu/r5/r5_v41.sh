#!/bin/bash
# round 5 visit 41: upper bound of what the split-K reduce launches of the weight gradients cost in the step (diagnostic build,
# SF_WGRAD_SKIP_REDUCE=1: the reduce kernels are not launched; gradients are garbage, timing only)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v41; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
export SFAMD_LIBRARY=$PWD/slowfast_amd/libsfamd_diag.so
ROUNDS=2 bash tools/gpu/ab.sh $D -- "slowfast diag:X=1" "slowfast no reduce:SF_WGRAD_SKIP_REDUCE=1"
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit diag:X=1" "mvit no reduce:SF_WGRAD_SKIP_REDUCE=1"
echo "exit 0"
