#!/bin/bash
# round 5 visit 28: software-pipelined GEMM store loops (loads of row k + 1 before the store of row k), attention forward at three
# workgroups per CU
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v28; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 1500 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_kernels_gpu.py > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $D/pytest.log | cut -c1-300
timeout 300 python tools/token_bench.py --only attn --iters 20 2>&1 | grep -v amdgpu.ids | tee $D/token_bench.txt
timeout 300 python tools/gemm_cold_bench.py 2>&1 | grep -v amdgpu.ids | tee $D/cold.txt
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit:X=1"
ROUNDS=2 bash tools/gpu/ab.sh $D -- "slowfast:X=1"
echo "exit 0"
