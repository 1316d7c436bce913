#!/bin/bash
# round 5 visit 31: in-step A/B of the working tree against the previous commit's library (slowfast_amd/libsfamd_prev.so, built by
# hand from `git show HEAD:...`; SF_ALLOW_STALE_LIBRARY=1 lets the loader take it)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v31; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
ROUNDS=3 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit new:X=1" "mvit prev:SFAMD_LIBRARY=$PWD/slowfast_amd/libsfamd_prev.so,SF_ALLOW_STALE_LIBRARY=1"
echo "exit 0"
