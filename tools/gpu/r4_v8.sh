#!/bin/bash
# round 4 visit 8: key-side attention backward with DMA'd rq rows + one barrier per chunk, query split target (SF_ATTN_DKV_WGS),
# forward-kernel ablation (64: K / V / OH staged once, 128: staging only), parity, in-step A/B.
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v8; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py -k "attention" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
SF_ATTN_DKV_WGS=512 timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_model_gpu.py -k "attention or mvit_matches or MVIT" > $D/pytest512.log 2>&1; echo "pytest wgs512 rc=$?"; tail -2 $D/pytest512.log | cut -c1-300
: > $D/attn.txt
for W in 1024 512 256; do for K in 2 1; do
  SF_ATTN_DKV_WGS=$W SF_ATTN_DKV_KT=$K timeout 120 python tools/token_bench.py --iters 20 --only attn 2>&1 | grep "^attn" | sed "s/^/wgs=$W kt=$K /" | tee -a $D/attn.txt
done; done
for A in 64 128 192; do
  SF_ATTN_ABLATE=$A timeout 120 python tools/token_bench.py --iters 20 --only stage3attn 2>&1 | grep "^attn" | sed "s/^/fwd ablate=$A /" | tee -a $D/attn.txt
done
B="--preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
for R in 1 2; do for W in 1024 512; do
  SF_ATTN_DKV_WGS=$W timeout 300 python bench.py $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit dkv_wgs=$W', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
done; done
echo "exit 0"
