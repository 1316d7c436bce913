#!/bin/bash
# GPU visit 27: X3D training / eval tests after the norm-container refactor, TrainStep tests, X3D-M eval throughput.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_model_gpu.py tests/test_step.py -q --tb=short -k "x3d or test_step or graph" > gpurun_out/pytest_gpu27.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu27.log | tail -8 | cut -c1-400
timeout 60 python tools/bench_eval.py --preset X3D_M --batch 64 --steps 5 > gpurun_out/bench_eval_x3d.log 2>&1; echo "bench_eval rc=$?"; tail -1 gpurun_out/bench_eval_x3d.log | cut -c1-1200
