#!/bin/bash
# round 4 visit 14: the rel-pos scatter zero-fills E itself (no memset node in the captured graph) -- probe, full GPU suite, bench
D=gpurun_out/v14; mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for M in graph eager; do PROBE_DIRTY=0 timeout 300 python tools/gpu/r4_nan_probe.py $M 5 2>&1 | grep -E "^(graph|eager) it|Error|error" | cut -c1-300; done > $D/probe.txt 2>&1
cat $D/probe.txt
if grep -q "nonfinite params [1-9]" $D/probe.txt; then echo "STILL NON-FINITE"; exit 0; fi
rm -f $D/parity_report.jsonl
SF_PARITY_REPORT=$R/$D/parity_report.jsonl timeout 1800 python -m pytest tests -q -m gpu --tb=short > $D/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; grep -E "passed|failed|FAILED|Error" $D/pytest_gpu.log | head -20 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $D/smoke.log | cut -c1-300
timeout 600 python bench.py > $D/bench.log 2> $D/bench.err; echo "bench rc=$?"; tail -1 $D/bench.log | cut -c1-400
timeout 300 python bench.py --preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > $D/bench_mvit.log 2> $D/bench_mvit.err; echo "bench mvit rc=$?"; tail -1 $D/bench_mvit.log | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/prof_mvit -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary --preset MVITv2_S_16x4 --batch 32 > $R/$D/rocprof_mvit.log 2>&1; echo "rocprof rc=$?"
cd $R
F=$(find $D/prof_mvit -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" $D/r4_v14_mvit_kernel_stats.md "round 4 visit 14: MViTv2-S bench command (3 timed + 2 warm-up steps) with the self-zeroing rel-pos scatter" > /dev/null 2>&1
grep -E "relpos|fillBuffer|total kernel" $D/r4_v14_mvit_kernel_stats.md
