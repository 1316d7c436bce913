#!/bin/bash
# round 5 visit 50: token / MViT-family GPU tests at HEAD (after the LayerNorm forward change)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v50; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 240 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_model_gpu.py tests/test_zy_new_families_gpu.py -k "tokens or mvit or vit or rev or layernorm" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-200
echo "exit 0"
