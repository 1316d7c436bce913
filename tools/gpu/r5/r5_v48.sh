#!/bin/bash
# round 5 visit 48: rel-pos gather through an LDS image of 32 rows of G (sf_relpos_gather_lds_kernel) on / off
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v48; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
ROUNDS=3 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit lds gather:X=1" "mvit off:SF_RELPOS_GA_LDS=0"
echo "exit 0"
