#!/bin/bash
# round 5 visit 17: MViT q / kv pooling chains on two streams (engine.run_branches)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v17; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_step.py -k "branch_streams or mvit" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $D/pytest.log | cut -c1-300
ROUNDS=3 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit branches=1:SF_BRANCH_STREAMS=1" "mvit branches=0:SF_BRANCH_STREAMS=0"
ROUNDS=1 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 --no-graph -- "mvit eager branches=1:SF_BRANCH_STREAMS=1" "mvit eager branches=0:SF_BRANCH_STREAMS=0"
echo "exit 0"
