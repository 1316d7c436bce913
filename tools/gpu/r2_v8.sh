#!/bin/bash
# round 2, GPU visit 8: regenerate the autocast yardstick (GradScaler back-off, eval cases), FULL gpu suite with the parity report,
# default bench line (SlowFast + MViTv2-S secondary + cpu baseline).
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 420 python tools/autocast_yardstick.py --out gpurun_out/autocast_yardstick.json > gpurun_out/autocast8.log 2>&1; echo "autocast rc=$?"; grep -c "ERROR" gpurun_out/autocast8.log
cp gpurun_out/autocast_yardstick.json tests/golden/autocast_yardstick.json
rm -f gpurun_out/parity_report.jsonl
SF_PARITY_REPORT=$PWD/gpurun_out/parity_report.jsonl timeout 900 python -m pytest tests -q -m gpu --tb=short -x > gpurun_out/pytest8.log 2>&1; echo "pytest gpu rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest8.log | tail -8 | cut -c1-400
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke8.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke8.log | cut -c1-300
timeout 400 python bench.py > gpurun_out/bench8.log 2> gpurun_out/bench8.err; echo "bench rc=$?"; tail -1 gpurun_out/bench8.log | cut -c1-1500
