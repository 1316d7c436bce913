#!/bin/bash
# round 5 visit 30: attention backward kernels without the in-loop pipeline drains (lse / v / one-hot operands consumed before the
# loop, side loads unconditional, no spills: scalar-base chunk copies); f32 side rows early-out in the GEMM epilogues
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v30; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 1500 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_kernels_gpu.py > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $D/pytest.log | cut -c1-300
timeout 300 python tools/token_bench.py --only attn --iters 20 2>&1 | grep -v amdgpu.ids | tee $D/token_bench.txt
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit:X=1"
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_model_gpu.py -k "mvit or MVIT" > $D/pytest_model.log 2>&1; echo "pytest model rc=$?"; tail -3 $D/pytest_model.log | cut -c1-300
echo "exit 0"
