#!/bin/bash
# round 5 visit 34: partial-chunk clamp arithmetic kept inside the chunk loop (no spills in any 96-wide attention kernel) against HEAD
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v34; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit new:X=1" "mvit prev:SFAMD_LIBRARY=$PWD/slowfast_amd/libsfamd_prev.so,SF_ALLOW_STALE_LIBRARY=1"
echo "exit 0"
