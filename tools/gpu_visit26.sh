#!/bin/bash
# GPU visit 26: MViTv1 / ViT option family on the real kernels (fused + unfused attention), full-size MViTv1-B step.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_model_gpu.py -q --tb=short -k "mvit_v1_and_vit or mvit or sub_batchnorm" > gpurun_out/pytest_gpu26.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu26.log | tail -8 | cut -c1-400
timeout 100 python bench.py --preset MVIT_B_16x4_CONV --batch 32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_mvit_v1.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_mvit_v1.log | cut -c1-600
