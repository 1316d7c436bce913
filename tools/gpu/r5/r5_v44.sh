#!/bin/bash
# round 5 visit 44: query-split target of the key-side attention backward (workgroups per launch) 256 / 512 (default) / 1024, variant libraries
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v44; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
for W in 512 256 1024; do
  if [ $W = 512 ]; then E="X=1"; else E="SFAMD_LIBRARY=$PWD/slowfast_amd/libsfamd_wgs$W.so,SF_ALLOW_STALE_LIBRARY=1"; fi
  echo "== wgs $W" | tee -a $D/token_bench.txt
  env ${E//,/ } timeout 200 python tools/token_bench.py --only attn --iters 20 2>&1 | grep "^attn" | tee -a $D/token_bench.txt
done
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit wgs512:X=1" "mvit wgs256:SFAMD_LIBRARY=$PWD/slowfast_amd/libsfamd_wgs256.so,SF_ALLOW_STALE_LIBRARY=1" "mvit wgs1024:SFAMD_LIBRARY=$PWD/slowfast_amd/libsfamd_wgs1024.so,SF_ALLOW_STALE_LIBRARY=1"
echo "exit 0"
