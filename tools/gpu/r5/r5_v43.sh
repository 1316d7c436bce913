#!/bin/bash
# round 5 visit 43: deferred weight-gradient reductions flushed by slab bytes (cache residency) -- 32 / 96 / 256 MB / unlimited / off
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v43; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
ROUNDS=2 bash tools/gpu/ab.sh $D -- "slowfast 32MB:SF_DEFER_WGRAD_MB=32" "slowfast 96MB:X=1" "slowfast 256MB:SF_DEFER_WGRAD_MB=256" "slowfast unlimited:SF_DEFER_WGRAD_MB=100000" "slowfast off:SF_DEFER_WGRAD=0"
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit 32MB:SF_DEFER_WGRAD_MB=32" "mvit 96MB:X=1" "mvit 256MB:SF_DEFER_WGRAD_MB=256" "mvit off:SF_DEFER_WGRAD=0"
echo "exit 0"
