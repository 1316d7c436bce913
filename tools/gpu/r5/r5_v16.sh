#!/bin/bash
# round 5 visit 16: three-deep key-chunk ring in sf_attn_fwd / sf_attn_bwd_dq: parity, microbench, in-step
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v16; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py -k "attention" > $D/pytest.log 2>&1; echo "pytest attention rc=$?"; tail -3 $D/pytest.log | cut -c1-300
timeout 300 python tools/token_bench.py --iters 10 --only attn 2>&1 | grep -v amdgpu.ids | tee $D/token_bench.txt | tail -12 | cut -c1-200
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit ring3:X=1"
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_model_gpu.py -k "MVIT or mvit" > $D/pytest_model.log 2>&1; echo "pytest mvit model rc=$?"; tail -3 $D/pytest_model.log | cut -c1-300
echo "exit 0"
