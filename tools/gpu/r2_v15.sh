#!/bin/bash
# round 2, GPU visit 15: weight gradients on a side stream (fork / join inside the captured graphs): parity + A/B.
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 600 python -m pytest tests/test_step.py tests/test_model_gpu.py -q --tb=short -k "graph_replay or segmented or rccl or flat_optimizer or well_conditioned or blocks_strict or full_size_batch32" > gpurun_out/pytest15.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest15.log | tail -8 | cut -c1-400
for V in "side1 SF_WGRAD_STREAM=1" "side0 SF_WGRAD_STREAM=0" "side1b SF_WGRAD_STREAM=1" "side0b SF_WGRAD_STREAM=0"; do
  set -- $V
  env $2 timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-profile > gpurun_out/bench15_$1.log 2>&1; echo "bench $1 rc=$? $(tail -1 gpurun_out/bench15_$1.log | cut -c1-200)"
done
for P in "MVITv2_S_16x4 32 mvit" "X3D_M 64 x3d"; do
  set -- $P
  for S in 1 0; do
  SF_WGRAD_STREAM=$S timeout 150 python bench.py --preset $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench15_$3_$S.log 2>&1; echo "bench $3 side=$S rc=$? $(tail -1 gpurun_out/bench15_$3_$S.log | cut -c1-200)"
  done
done
