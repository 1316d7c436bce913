#!/bin/bash
# round 5 visit 49: LayerNorm forward with every row piece of a pass requested before any is converted (two rows really in flight)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v49; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
V="SFAMD_LIBRARY=$PWD/slowfast_amd/libsfamd_prev.so,SF_ALLOW_STALE_LIBRARY=1"
timeout 300 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py -k "layernorm or rows32" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $D/pytest.log | cut -c1-200
for L in "new:X=1" "prev:$V"; do
  echo "== ${L%%:*}" | tee -a $D/token_bench.txt
  E=${L#*:}; env ${E//,/ } timeout 200 python tools/token_bench.py --only ln --iters 20 2>&1 | grep "^layernorm" | tee -a $D/token_bench.txt
done
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit new:X=1" "mvit prev:$V"
echo "exit 0"
