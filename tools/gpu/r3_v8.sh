#!/bin/bash
# round 3 visit 8: bfloat16 storage mode (SF_ACT_DTYPE=bf16 -> libsfamd_bf16.so) on the GPU: the reference-derived yardstick
# (oracle graph under torch.autocast(bfloat16) on PyTorch-ROCm), kernel + model checks, one bench line.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/v8; export TMPDIR=/tmp
SF_ACT_DTYPE=bf16 timeout 900 python tools/autocast_yardstick.py --dtype bfloat16 --out gpurun_out/v8/autocast_yardstick_bf16.json \
   c2d_wc slowfast_wc x3d_wc r101nl_wc mvit_tiny slowfast_tiny c2d_tiny x3d_tiny > gpurun_out/v8/yardstick.log 2>&1
echo "yardstick rc=$?"; tail -3 gpurun_out/v8/yardstick.log
cp gpurun_out/v8/autocast_yardstick_bf16.json tests/golden/autocast_yardstick_bf16.json
SF_PARITY_REPORT=$PWD/gpurun_out/v8/parity.jsonl timeout 2400 python -m pytest tests/test_bf16_mode.py -x -q -m gpu -s > gpurun_out/v8/pytest_bf16.log 2>&1
echo "pytest bf16 rc=$?"; tail -12 gpurun_out/v8/pytest_bf16.log
for M in fp16 bf16; do
  SF_ACT_DTYPE=$M timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$M', d['dtype'], d['value'], d['ms_per_step'])"
done
SF_ACT_DTYPE=bf16 timeout 300 python bench.py --preset mvit --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit bf16', d['dtype'], d['value'], d['ms_per_step'])"
echo "exit 0"
