#!/bin/bash
# round 5 visit 37: max-pool backward with the four candidate windows requested together (sf_pool_bwd4_kernel) on / off
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v37; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_kernels_gpu.py tests/test_tokens_gpu.py -k "pool or gemm" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
ROUNDS=2 bash tools/gpu/ab.sh $D -- "slowfast bwd4:X=1" "slowfast off:SF_POOL_BWD4=0"
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit bwd4:X=1" "mvit off:SF_POOL_BWD4=0"
echo "exit 0"
