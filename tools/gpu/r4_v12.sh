#!/bin/bash
# round 4 visit 12: LayerNorm forward with two rows in flight (SF_LN_RU), mlp.fc2 / attn.proj bias gradients from norm2's backward
# pass (SF_LN_BIAS_SUMS): parity, in-step A/B
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v12; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_model_gpu.py -k "layernorm or rows32 or mvit_matches or mvit_drop or MVIT or resid_side" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
B="--preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
for R in 1 2; do for V in "1 0" "2 0" "1 1" "2 1"; do set -- $V
  SF_LN_RU=$1 SF_LN_BIAS_SUMS=$2 timeout 300 python bench.py $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit ln_ru=$1 ln_bias_sums=$2', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
done; done
echo "exit 0"
