#!/bin/bash
# round 3: the full -m gpu suite with the parity record + smoke + the default bench line (used when the full evidence script's
# profiles are already in place)
D=gpurun_out/final
mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f $D/parity_report.jsonl $D/parity_report_bf16.jsonl
SF_PARITY_REPORT=$R/$D/parity_report.jsonl timeout 2400 python -m pytest tests -x -q -m gpu --tb=short > $D/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; grep -E "passed|failed|FAILED|Error" $D/pytest_gpu.log | tail -4 | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $D/smoke.log | cut -c1-300
timeout 600 python bench.py > $D/bench2.log 2> $D/bench2.err; echo "bench rc=$?"; tail -1 $D/bench2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['secondary']['value'])"
echo "exit 0"
