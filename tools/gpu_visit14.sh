#!/bin/bash
# GPU visit 14: fused attention with two query tiles per wave (fwd, dq), dK/dV occupancy A/B, faster finalize kernels,
# RCCL single-rank TrainStep test, torchrun launch check; full GPU suite.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -12 | cut -c1-600
for V in "2 3 qt2" "1 3 qt1" "2 2 qt2occ2"; do
  set -- $V
  SF_ATTN_QT=$1 SF_ATTN_DKV_OCC=$2 timeout 600 python bench.py --preset MVITv2_S_16x4 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_mvit_$3.log 2>&1; echo "bench mvit $3 rc=$?"; tail -1 gpurun_out/bench_mvit_$3.log | cut -c1-1300
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_slowfast.log 2>&1; echo "bench slowfast rc=$?"; tail -1 gpurun_out/bench_slowfast.log | cut -c1-1300
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29537 bench.py --gpus 1 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_torchrun.log 2>&1; echo "bench torchrun rc=$?"; tail -1 gpurun_out/bench_torchrun.log | cut -c1-300
SF_ATTN_QT=2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v14_mvit -- python bench.py --preset MVITv2_S_16x4 --batch 32 --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof_mvit.log 2>&1; echo "rocprof mvit rc=$?"
