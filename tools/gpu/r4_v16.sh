#!/bin/bash
# round 4 visit 16: LayerNorm forward on C / 24 lanes x three slots (no idle lanes at C = 96 * 2^k): microbenchmark + in-step A/B
D=gpurun_out/v16; mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 600 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_model_gpu.py -k "layernorm or rows32 or mvit_matches or MVIT or resid_side" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
for V in "SF_LN_FWD_NS3=0" "SF_LN_FWD_NS3=1" "SF_LN_FWD_NS3=1 SF_LN_RU=1"; do
  echo "== $V" | tee -a $D/r4_v16_ln_bench.txt
  env $V timeout 200 python tools/token_bench.py --iters 20 --only ln 2>&1 | grep layernorm | tee -a $D/r4_v16_ln_bench.txt
done
for V in "SF_LN_FWD_NS3=0" "SF_LN_FWD_NS3=1" "SF_LN_FWD_NS3=0" "SF_LN_FWD_NS3=1"; do
  env $V timeout 300 python bench.py --preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit $V', d['value'], d['ms_per_step'])" | tee -a $D/r4_v16_ln_ab.txt
done
