#!/bin/bash
# evidence visit: full GPU suite, smoke, default bench line, kernel stats + queue timeline + HBM-traffic PMC passes of the
# bench command for both headline models, all presets.  Usage: ROUND=6 bash tools/gpu/final.sh [tag] [noprof]  ->  gpurun_out/r6_<tag>/ (files r6_<tag>_*)
cd "$GRAFT_REPO_ROOT"; TAG=${1:-final}; D=gpurun_out/r${ROUND:-6}_$TAG; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
rm -f $D/parity_report.jsonl
SF_PARITY_REPORT=$R/$D/parity_report.jsonl timeout 2400 python -m pytest tests -q -m gpu --tb=short > $D/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; grep -E "passed|failed|FAILED|Error" $D/pytest_gpu.log | tail -6 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $D/smoke.log | cut -c1-300
timeout 900 python bench.py > $D/bench.log 2> $D/bench.err; echo "bench rc=$?"; tail -1 $D/bench.log | cut -c1-600
if [ "$2" != "noprof" ]; then
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary"
for P in "SLOWFAST_8x8_R50 32 slowfast" "MVITv2_S_16x4 32 mvit" "X3D_M 64 x3d" "SLOWFAST_32x2_R101_50_50 16 r101"; do
  set -- $P
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/prof_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/$D/rocprof_$3.log 2>&1; echo "rocprof $3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$D/pmc_fetch_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/$D/pmc_fetch_$3.log 2>&1; echo "pmc fetch $3 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$D/pmc_write_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/$D/pmc_write_$3.log 2>&1; echo "pmc write $3 rc=$?"
done
cd $R
for n in slowfast mvit x3d r101; do
  F=$(find $D/prof_$n -name "*kernel_stats.csv" | head -1)
  python tools/rocprof_summary.py "$F" $D/r${ROUND:-6}_${TAG}_${n}_kernel_stats.md "round ${ROUND:-6} ($TAG, HEAD): $n bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
  T=$(find $D/prof_$n -name "*kernel_trace.csv" | head -1)
  python tools/stream_timeline.py "$T" $D/r${ROUND:-6}_${TAG}_${n}_timeline.md > /dev/null 2>&1
  FF=$(find $D/pmc_fetch_$n -name "*counter_collection.csv" | head -1); FW=$(find $D/pmc_write_$n -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py "$FF" "$FW" $D/pmc_traffic_$n.json "$(python -c 'from slowfast_amd import lib; print(lib.get_lib().build_id)')" > $D/pmc_traffic_$n.txt 2>&1
  head -12 $D/r${ROUND:-6}_${TAG}_${n}_kernel_stats.md | tail -5 | cut -c1-160
  head -12 $D/r${ROUND:-6}_${TAG}_${n}_timeline.md | cut -c1-160
  rm -rf $D/prof_$n $D/pmc_fetch_$n $D/pmc_write_$n
done
timeout 900 bash tools/gpu/all_presets.sh > $D/presets.log 2>&1; echo "presets rc=$?"; cut -c1-200 $D/presets.log
python tools/collect_presets.py gpurun_out/presets $D/presets.json
fi
echo "exit 0"
