#!/bin/bash
# round 5 visit 39: key-side attention backward with ONE key tile per wave at three waves per SIMD (variant library) against two tiles at two
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v39; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
V="SFAMD_LIBRARY=$PWD/slowfast_amd/libsfamd_prev.so,SF_ALLOW_STALE_LIBRARY=1"
for L in "kt2 occ2:X=1" "kt1 occ3:$V"; do
  echo "== ${L%%:*}" | tee -a $D/token_bench.txt
  E=${L#*:}; env ${E//,/ } timeout 200 python tools/token_bench.py --only attn --iters 20 2>&1 | grep "^attn" | tee -a $D/token_bench.txt
done
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit kt2 occ2:X=1" "mvit kt1 occ3:$V"
echo "exit 0"
