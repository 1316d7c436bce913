#!/bin/bash
# round 4 visit 20: sf_colsum with four rows in flight against the previous binary (tools/gpu/ab/libsfamd_old.so)
D=gpurun_out/v20; mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OLD="SFAMD_LIBRARY=$R/tools/gpu/ab/libsfamd_old.so SF_ALLOW_STALE_LIBRARY=1"
for V in old new; do
  E="SF_NOOP=1"; [ $V = old ] && E="$OLD"
  echo "== $V" | tee -a $D/r4_v20_colsum_bench.txt
  env $E timeout 200 python tools/token_bench.py --iters 20 --only colsum 2>&1 | grep colsum | tee -a $D/r4_v20_colsum_bench.txt
done
for V in old new old new; do
  E="SF_NOOP=1"; [ $V = old ] && E="$OLD"
  env $E timeout 300 python bench.py --preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit colsum $V', d['value'], d['ms_per_step'])" | tee -a $D/r4_v20_colsum_ab.txt
done
timeout 600 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py -k "bias or colsum or gemm or linear" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
