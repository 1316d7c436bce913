#!/bin/bash
# round 3, visit 1: the new parity tests on hardware (full -m gpu suite with the parity report) + the unchanged-kernel baseline bench.
mkdir -p gpurun_out/v1
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/v1/parity_report.jsonl
SF_PARITY_REPORT=$R/gpurun_out/v1/parity_report.jsonl timeout 1500 python -m pytest tests -q -m gpu --tb=short -s > gpurun_out/v1/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/v1/pytest_gpu.log | tail -12 | cut -c1-400
timeout 400 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/v1/bench.log 2> gpurun_out/v1/bench.err; echo "bench rc=$?"; tail -1 gpurun_out/v1/bench.log | cut -c1-600
