#!/bin/bash
# round 2, GPU visit 12: re-run the step tests after the packed-weight cache fix.
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 600 python -m pytest tests/test_step.py -q --tb=short > gpurun_out/pytest12.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest12.log | tail -8 | cut -c1-400
