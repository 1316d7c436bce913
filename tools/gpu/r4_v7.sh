#!/bin/bash
# round 4 visit 7: what does the key-side attention backward kernel wait for?  Parts switched off one at a time (SF_ATTN_ABLATE,
# results garbage) on the stage-3 shape, warm: time of the whole backward call (dq + dkv + reduce) -- only the dkv kernel changes.
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v7; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
: > $D/ablate.txt
for A in 0 1 2 4 8 16 32 3 5 6 7 24 56 63 39 47; do
  SF_ATTN_ABLATE=$A timeout 120 python tools/token_bench.py --iters 20 --only stage3attn 2>&1 | grep "^attn" | sed "s/^/ablate=$A /" | tee -a $D/ablate.txt
done
for A in 0 7 63; do
  SF_ATTN_DKV_KT=1 SF_ATTN_ABLATE=$A timeout 120 python tools/token_bench.py --iters 20 --only stage3attn 2>&1 | grep "^attn" | sed "s/^/kt1 ablate=$A /" | tee -a $D/ablate.txt
done
cd /tmp
for A in 0 63; do
SF_ATTN_ABLATE=$A timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/prof$A -o p -- python $GRAFT_REPO_ROOT/tools/token_bench.py --iters 10 --only stage3attn > /dev/null 2>&1
F=$(find $GRAFT_REPO_ROOT/$D/prof$A -name "*kernel_stats.csv" | head -1); echo "== ablate=$A"; grep -i "attn" "$F" | cut -d, -f1-4 | cut -c1-140
done
cd $GRAFT_REPO_ROOT; find $D -name "*.csv" -size +1M -delete
echo "exit 0"
