#!/bin/bash
# GPU visit 25: fused thin-stem epilogue (bias / ReLU in sf_stem_fwd_kernel), eval-path throughput with a per-entry profile.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_kernels_gpu.py::test_conv_fwd_fused "tests/test_model_gpu.py::test_eval_path_matches_reference" \
  "tests/test_kernels_gpu.py" -q --tb=short -k "fused or stem or eval_slowfast_tiny" > gpurun_out/pytest_gpu25.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu25.log | tail -8 | cut -c1-400
timeout 80 python tools/bench_eval.py --steps 5 > gpurun_out/bench_eval25.log 2>&1; echo "bench_eval rc=$?"; tail -1 gpurun_out/bench_eval25.log | cut -c1-1500
