#!/bin/bash
# round 2, final evidence visit (after the launch-count cuts): full gpu suite, smoke, default bench line, rocprofv3 kernel stats + PMC passes of the same command.
mkdir -p gpurun_out/final4
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/final4/parity_report.jsonl
SF_PARITY_REPORT=$R/gpurun_out/final4/parity_report.jsonl timeout 1200 python -m pytest tests -x -q -m gpu --tb=short > gpurun_out/final4/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/final4/pytest_gpu.log | tail -6 | cut -c1-300
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final4/smoke.log 2>&1; echo "smoke rc=$?"
timeout 500 python bench.py > gpurun_out/final4/bench.log 2> gpurun_out/final4/bench.err; echo "bench rc=$?"; tail -1 gpurun_out/final4/bench.log | cut -c1-400
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary"
for P in "SLOWFAST_8x8_R50 32 slowfast" "MVITv2_S_16x4 32 mvit"; do
  set -- $P
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final4/prof_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/gpurun_out/final4/rocprof_$3.log 2>&1; echo "rocprof $3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final4/pmc_fetch_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/gpurun_out/final4/pmc_fetch_$3.log 2>&1; echo "pmc fetch $3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final4/pmc_write_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/gpurun_out/final4/pmc_write_$3.log 2>&1; echo "pmc write $3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/final4/pmc_mfma_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/gpurun_out/final4/pmc_mfma_$3.log 2>&1; echo "pmc mfma $3 rc=$?"
done
cd $R
for n in slowfast mvit; do
  F=$(find gpurun_out/final4/prof_$n -name "*kernel_stats.csv" | head -1)
  python tools/rocprof_summary.py "$F" gpurun_out/final4/r2_final_${n}_kernel_stats.md "round 2 final: $n default bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
  FF=$(find gpurun_out/final4/pmc_fetch_$n -name "*counter_collection.csv" | head -1); FW=$(find gpurun_out/final4/pmc_write_$n -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py "$FF" "$FW" gpurun_out/final4/pmc_traffic_$n.json > gpurun_out/final4/pmc_traffic_$n.txt 2>&1
  FM=$(find gpurun_out/final4/pmc_mfma_$n -name "*counter_collection.csv" | head -1)
  python tools/pmc_metric.py gpurun_out/final4/r2_final_pmc_mfma_$n.md "round 2 final: MFMA / wave-state counters, $n bench" "$FM" > /dev/null 2>&1
  head -16 gpurun_out/final4/r2_final_${n}_kernel_stats.md | tail -9 | cut -c1-160
done
find gpurun_out/final4 -name "*.csv" -size +1M -delete
