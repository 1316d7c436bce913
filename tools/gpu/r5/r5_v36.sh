#!/bin/bash
# round 5 visit 36: 128 x 128 igemm2 tiles as the default for eligible GEMMs: SlowFast / X3D / MViT in-step on/off, GPU kernel tests
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v36; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 1500 python -m pytest -q -m gpu -x --tb=short tests/test_kernels_gpu.py tests/test_tokens_gpu.py > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
ROUNDS=2 bash tools/gpu/ab.sh $D -- "slowfast t128:X=1" "slowfast off:SF_IGEMM2_T128=0"
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit t128:X=1" "mvit off:SF_IGEMM2_T128=0"
echo "exit 0"
