#!/bin/bash
# round 2, GPU visit 11: segmented graph replay + FlatOptimizer on hardware; bench A/B (segmented / torch optimizer).
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 600 python -m pytest tests/test_step.py tests/test_model_gpu.py -q --tb=short -k "segmented or flat_optimizer or graph_replay or rccl or full_size_batch2" > gpurun_out/pytest11.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest11.log | tail -8 | cut -c1-400
timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/bench11_flat.log 2>&1; echo "bench flat rc=$? $(tail -1 gpurun_out/bench11_flat.log | cut -c1-200)"
timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --torch-optimizer --no-kernel-profile > gpurun_out/bench11_torch.log 2>&1; echo "bench torch-opt rc=$? $(tail -1 gpurun_out/bench11_torch.log | cut -c1-200)"
