#!/bin/bash
# round 4 visit 1: fp32 side rows of the MViT residual stream -- kernel checks, the engine-level consistency check, the full-size
# parity of BASELINE config 4 at the flat 1e-3 bound, the packed-loader test; cost of the side rows (A/B) and a SlowFast line.
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v1; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
rm -f $D/parity.jsonl
SF_PARITY_REPORT=$PWD/$D/parity.jsonl timeout 900 python -m pytest -q -m gpu -x --tb=short \
  tests/test_tokens_gpu.py::test_rows32_side_rows tests/test_model_gpu.py::test_mvit_resid_side_rows \
  "tests/test_model_gpu.py::test_full_size_batch2_against_oracle[MVITv2_S_16x4]" tests/test_step.py::test_packed_loader_writes_static_inputs \
  tests/test_tokens_gpu.py::test_gemm_linear tests/test_tokens_gpu.py::test_layernorm > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $D/pytest.log | cut -c1-400
cat $D/parity.jsonl | cut -c1-700
# the same case with class-token rows only / with the side rows off (what the logits read before)
for V in cls 0; do
SF_MVIT_RESID32=$V SF_PARITY_REPORT=$PWD/$D/parity_$V.jsonl timeout 600 python -m pytest -q -m gpu -x --tb=line \
  "tests/test_model_gpu.py::test_full_size_batch2_against_oracle[MVITv2_S_16x4]" > $D/pytest_$V.log 2>&1; echo "pytest (side rows $V) rc=$?"; cut -c1-200 $D/parity_$V.jsonl
done
B="python bench.py --preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
for R in 1 2; do for V in 0 cls 1; do
  SF_MVIT_RESID32=$V timeout 300 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit resid32=$V', d['value'], d['ms_per_step'])" | tee -a $D/mvit_ab.txt
done; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('slowfast', d['value'], d['ms_per_step'])" | tee $D/slowfast.txt
echo "exit 0"
