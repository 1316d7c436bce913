#!/bin/bash
# round 4 visit 24: tiled depthwise kernels with vector-indexed (ds_write_b128) staging stores and the weight gradient's sliding
# window unrolled over its rotation, against the previous binary (tools/gpu/ab/libsfamd_old.so); then the full GPU suite
D=gpurun_out/v24; mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OLD="SFAMD_LIBRARY=$R/tools/gpu/ab/libsfamd_old.so SF_ALLOW_STALE_LIBRARY=1"
for V in old new; do
  E="SF_NOOP=1"; [ $V = old ] && E="$OLD"
  echo "== $V" | tee -a $D/r4_v24_dw_bench.txt
  env $E timeout 200 python tools/token_bench.py --iters 20 --only stage3 2>&1 | grep -E "^dwconv" | tee -a $D/r4_v24_dw_bench.txt
done
for V in old new old new; do
  E="SF_NOOP=1"; [ $V = old ] && E="$OLD"
  env $E timeout 300 python bench.py --preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit dwtile $V', d['value'], d['ms_per_step'])" | tee -a $D/r4_v24_dw_ab.txt
done
rm -f $D/parity_report.jsonl
SF_PARITY_REPORT=$R/$D/parity_report.jsonl timeout 1500 python -m pytest tests -q -m gpu --tb=short > $D/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; grep -E "passed|failed|FAILED|Error" $D/pytest_gpu.log | tail -6 | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $D/smoke.log | cut -c1-200
timeout 600 python bench.py > $D/bench.log 2> $D/bench.err; echo "bench rc=$?"; tail -1 $D/bench.log | cut -c1-300
