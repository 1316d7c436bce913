#!/bin/bash
# round 5 visit 47: rel-pos gradient scatter through an LDS image of the chunk (sf_relpos_scatter_lds_kernel) on / off
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v47; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
ROUNDS=3 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit lds scatter:X=1" "mvit off:SF_RELPOS_SC_LDS=0"
echo "exit 0"
