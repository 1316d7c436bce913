#!/bin/bash
# round 4 visit 22: graph replay with poisoned allocations (new GPU test), tests/test_step.py as a whole
D=gpurun_out/v22; mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu --tb=short tests/test_step.py > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $D/pytest.log | cut -c1-300
