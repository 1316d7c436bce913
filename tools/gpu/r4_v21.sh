#!/bin/bash
# round 4 visit 21: GELU / GELU' in the GEMM epilogues from one v_rcp + one v_exp + 9 FMAs (A&S 7.1.26 erfc) against ocml
# erff + expf (tools/gpu/ab/libsfamd_old.so): MViT step A/B, then parity (token kernels, MViT models incl. the full-size case)
D=gpurun_out/v21; mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OLD="SFAMD_LIBRARY=$R/tools/gpu/ab/libsfamd_old.so SF_ALLOW_STALE_LIBRARY=1"
for V in old new old new; do
  E="SF_NOOP=1"; [ $V = old ] && E="$OLD"
  env $E timeout 300 python bench.py --preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit gelu $V', d['value'], d['ms_per_step'])" | tee -a $D/r4_v21_gelu_ab.txt
done
SF_PARITY_REPORT=$R/$D/parity.jsonl timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_model_gpu.py -k "gelu or gemm or mlp or mvit or MVIT or vit" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
grep -o '"case": "MVITv2_S_16x4@full"[^}]*"logits_max": [0-9.e-]*' $D/parity.jsonl | cut -c1-200
