#!/bin/bash
# round 4, second final evidence visit (HEAD after the LayerNorm-backward, rel-pos scatter and tiled depthwise weight-gradient
# changes): full gpu suite with the parity record, smoke, default bench line, switches on / off on this box, token
# microbenchmarks, bf16 lines, rocprofv3 kernel stats + HBM-traffic PMC passes of the bench command for both models.
D=gpurun_out/final2
mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f $D/parity_report.jsonl
SF_PARITY_REPORT=$R/$D/parity_report.jsonl timeout 1800 python -m pytest tests -q -m gpu --tb=short > $D/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; grep -E "passed|failed|FAILED|Error" $D/pytest_gpu.log | tail -6 | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $D/smoke.log | cut -c1-300
timeout 600 python bench.py > $D/bench.log 2> $D/bench.err; echo "bench rc=$?"; tail -1 $D/bench.log | cut -c1-400
OFF="SF_ATTN_DKV_KT=1 SF_ATTN_DKV_WGS=1024 SF_FIN_BATCH=0 SF_DW_XCD=0 SF_DW_TILED=0 SF_DW_WGRAD_TILED=0 SF_MVIT_RESID32=0 SF_LN_RU=1 SF_LN_BWD_RU=1 SF_LN_FWD_BLOCKS=4096 SF_LN_BIAS_SUMS=0 SF_STEM_XCD=0"
for P in SLOWFAST_8x8_R50 MVITv2_S_16x4; do
  for V in on off on off; do
    E="SF_NOOP=1"; [ $V = off ] && E="$OFF"
    env $E timeout 300 python bench.py --preset $P --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$P round-4 switches $V', d['value'], d['ms_per_step'])" | tee -a $D/r4_final2_switches_ab.txt
  done
done
timeout 300 python tools/token_bench.py --iters 10 2>&1 | grep -v amdgpu.ids > $D/r4_final2_token_bench.txt; tail -4 $D/r4_final2_token_bench.txt | cut -c1-160
SF_ACT_DTYPE=bf16 timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2> /dev/null | tail -1 > $D/bench_slowfast_bf16.json; python -c "import json; d=json.load(open('$D/bench_slowfast_bf16.json')); print('slowfast bf16', d['dtype'], d['value'], d['ms_per_step'])"
SF_ACT_DTYPE=bf16 timeout 300 python bench.py --preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2> /dev/null | tail -1 > $D/bench_mvit_bf16.json; python -c "import json; d=json.load(open('$D/bench_mvit_bf16.json')); print('mvit bf16', d['dtype'], d['value'], d['ms_per_step'])"
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary"
for P in "SLOWFAST_8x8_R50 32 slowfast" "MVITv2_S_16x4 32 mvit"; do
  set -- $P
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/prof_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/$D/rocprof_$3.log 2>&1; echo "rocprof $3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$D/pmc_fetch_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/$D/pmc_fetch_$3.log 2>&1; echo "pmc fetch $3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$D/pmc_write_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/$D/pmc_write_$3.log 2>&1; echo "pmc write $3 rc=$?"
done
cd $R
for n in slowfast mvit; do
  F=$(find $D/prof_$n -name "*kernel_stats.csv" | head -1)
  python tools/rocprof_summary.py "$F" $D/r4_final2_${n}_kernel_stats.md "round 4 final (second visit, HEAD): $n default bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
  FF=$(find $D/pmc_fetch_$n -name "*counter_collection.csv" | head -1); FW=$(find $D/pmc_write_$n -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py "$FF" "$FW" $D/pmc_traffic_$n.json > $D/pmc_traffic_$n.txt 2>&1
  head -14 $D/r4_final2_${n}_kernel_stats.md | tail -7 | cut -c1-160
done
