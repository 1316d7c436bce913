#!/bin/bash
# round 4, final evidence visit: full gpu suite (with the parity record), smoke, default bench line, rocprofv3 kernel stats + PMC
# passes (HBM traffic, MFMA / wave state) of the same command, the weight-gradient tile sweep, one bf16 bench line per model.
D=gpurun_out/final
mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f $D/parity_report.jsonl $D/parity_report_bf16.jsonl
SF_PARITY_REPORT=$R/$D/parity_report.jsonl timeout 1800 python -m pytest tests -x -q -m gpu --tb=short > $D/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; grep -E "passed|failed|FAILED|Error" $D/pytest_gpu.log | tail -4 | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $D/smoke.log | cut -c1-300
timeout 600 python bench.py > $D/bench.log 2> $D/bench.err; echo "bench rc=$?"; tail -1 $D/bench.log | cut -c1-600
# the same two models with every round-4 switch off (the round-3 schedule on this box): what the round changed, same box
OFF="SF_ATTN_DKV_KT=1 SF_ATTN_DKV_WGS=1024 SF_FIN_BATCH=0 SF_DW_XCD=0 SF_DW_TILED=0 SF_MVIT_RESID32=0 SF_LN_RU=1 SF_LN_BIAS_SUMS=0 SF_STEM_XCD=0"
for P in SLOWFAST_8x8_R50 MVITv2_S_16x4; do
  for V in on off; do
    E=""; [ $V = off ] && E="$OFF"
    env $E timeout 300 python bench.py --preset $P --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$P round-4 switches $V', d['value'], d['ms_per_step'])" | tee -a $D/r4_final_switches_ab.txt
  done
done
timeout 200 python tools/token_bench.py --iters 10 2>&1 | grep -v amdgpu.ids > $D/r4_final_token_bench.txt; tail -3 $D/r4_final_token_bench.txt | cut -c1-160
for M in bf16; do
  SF_ACT_DTYPE=$M timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2> /dev/null | tail -1 > $D/bench_slowfast_$M.json; python -c "import json; d=json.load(open('$D/bench_slowfast_$M.json')); print('slowfast $M', d['dtype'], d['value'], d['ms_per_step'])"
  SF_ACT_DTYPE=$M timeout 300 python bench.py --preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2> /dev/null | tail -1 > $D/bench_mvit_$M.json; python -c "import json; d=json.load(open('$D/bench_mvit_$M.json')); print('mvit $M', d['dtype'], d['value'], d['ms_per_step'])"
done
timeout 300 python tools/microbench.py --md $D/r4_final_per_geometry.md > $D/microbench.log 2>&1; echo "microbench rc=$?"; tail -1 $D/microbench.log | cut -c1-200
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary"
for P in "SLOWFAST_8x8_R50 32 slowfast" "MVITv2_S_16x4 32 mvit"; do
  set -- $P
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/prof_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/$D/rocprof_$3.log 2>&1; echo "rocprof $3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$D/pmc_fetch_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/$D/pmc_fetch_$3.log 2>&1; echo "pmc fetch $3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$D/pmc_write_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/$D/pmc_write_$3.log 2>&1; echo "pmc write $3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/$D/pmc_mfma_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/$D/pmc_mfma_$3.log 2>&1; echo "pmc mfma $3 rc=$?"
done
cd $R
for n in slowfast mvit; do
  F=$(find $D/prof_$n -name "*kernel_stats.csv" | head -1)
  python tools/rocprof_summary.py "$F" $D/r4_final_${n}_kernel_stats.md "round 4 final: $n default bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
  T=$(find $D/prof_$n -name "*kernel_trace.csv" | head -1)
  python tools/trace_neighbors.py "$T" copyBuffer > $D/r4_final_copybuffer_$n.txt 2>&1
  FF=$(find $D/pmc_fetch_$n -name "*counter_collection.csv" | head -1); FW=$(find $D/pmc_write_$n -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py "$FF" "$FW" $D/pmc_traffic_$n.json > $D/pmc_traffic_$n.txt 2>&1
  FM=$(find $D/pmc_mfma_$n -name "*counter_collection.csv" | head -1)
  python tools/pmc_metric.py $D/r4_final_pmc_mfma_$n.md "round 4 final: MFMA / wave-state counters, $n bench" "$FM" > /dev/null 2>&1
  head -16 $D/r4_final_${n}_kernel_stats.md | tail -9 | cut -c1-160
done
find $D -name "*.csv" -size +1M -delete
echo "exit 0"
