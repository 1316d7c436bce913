#!/bin/bash
# round 5 visit 26: attention query-side kernels -- prologue / epilogue round trips batched, lazy cross-lane maximum
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v26; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 1200 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_model_gpu.py -k "tokens or mvit or attn or attention" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $D/pytest.log | cut -c1-300
timeout 300 python tools/token_bench.py --only attn --iters 20 2>&1 | grep -v amdgpu.ids | tee $D/token_bench.txt
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit:X=1"
echo "exit 0"
