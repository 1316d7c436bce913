#!/bin/bash
# GPU visit 24 (7 GPU-minutes left in the round): the new eval-path kernels / tests, the re-conditioned NONLOCAL.GROUP case,
# SubBatchNorm, eval-path throughput, and a short default bench.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 170 python -m pytest tests/test_kernels_gpu.py::test_conv_fwd_fused "tests/test_model_gpu.py::test_nonlocal_matches_reference[slowfast_nln_group_tiny]" \
  tests/test_model_gpu.py::test_eval_path_matches_reference tests/test_model_gpu.py::test_eval_path_x3d_mvit \
  tests/test_model_gpu.py::test_test_step_graph_replay_equals_eager tests/test_model_gpu.py::test_sub_batchnorm_matches_reference \
  "tests/test_model_gpu.py::test_tiny_wiring[c2d_tiny]" -q --tb=short -s > gpurun_out/pytest_gpu24.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu24.log | tail -12 | cut -c1-400
timeout 60 python tools/bench_eval.py --steps 5 > gpurun_out/bench_eval.log 2>&1; echo "bench_eval rc=$?"; tail -1 gpurun_out/bench_eval.log | cut -c1-400
timeout 80 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_default24.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_default24.log | cut -c1-330
