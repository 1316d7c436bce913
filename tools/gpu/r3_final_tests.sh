#!/bin/bash
# round 3: the full -m gpu suite with the parity record (the first final visit stopped at a broken import in a test helper)
D=gpurun_out/final
mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f $D/parity_report.jsonl $D/parity_report_bf16.jsonl
SF_PARITY_REPORT=$R/$D/parity_report.jsonl timeout 2400 python -m pytest tests -x -q -m gpu --tb=short > $D/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; grep -E "passed|failed|FAILED|Error" $D/pytest_gpu.log | tail -4 | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $D/smoke.log | cut -c1-300
echo "exit 0"
