#!/bin/bash
# round 5 visit 27: what the attention forward kernel's time is made of (diagnostic build, parts switched off; results are garbage)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v27; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
export SFAMD_LIBRARY=$PWD/slowfast_amd/libsfamd_diag.so
for A in 0 64 128 192 256 512 768 448 704 960 1984 2048; do
  echo -n "ablate=$A " | tee -a $D/fwd_ablation.txt
  SF_ATTN_ABLATE=$A timeout 120 python tools/token_bench.py --only stage3attn --iters 30 2>&1 | grep "^attn" | tee -a $D/fwd_ablation.txt
done
for Q in 1 2; do
  echo -n "qt=$Q " | tee -a $D/fwd_ablation.txt
  SF_ATTN_QT=$Q timeout 120 python tools/token_bench.py --only stage3attn --iters 30 2>&1 | grep "^attn" | tee -a $D/fwd_ablation.txt
done
echo "exit 0"
