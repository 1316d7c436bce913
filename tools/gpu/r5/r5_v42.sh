#!/bin/bash
# round 5 visit 42: split reductions of the weight gradients queued per backward segment and run in batches (ops.deferred_wgrads) on / off
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v42; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 1500 python -m pytest -q -m gpu -x --tb=short tests/test_kernels_gpu.py tests/test_step.py > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
ROUNDS=2 bash tools/gpu/ab.sh $D -- "slowfast deferred:X=1" "slowfast off:SF_DEFER_WGRAD=0"
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit deferred:X=1" "mvit off:SF_DEFER_WGRAD=0"
echo "exit 0"
