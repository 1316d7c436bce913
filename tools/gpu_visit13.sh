#!/bin/bash
# GPU visit 13: fused attention v2 (bias / bias gradient on MFMA, exp2, lazy rescale, query-split dK/dV): parity + A/B.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --tb=short -k "attention or abi or mvit or tokens" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu (attention subset) rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -12 | cut -c1-600
SF_ATTN_FUSED=1 timeout 600 python bench.py --preset MVITv2_S_16x4 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_mvit_fused.log 2>&1; echo "bench mvit fused rc=$?"; tail -1 gpurun_out/bench_mvit_fused.log | cut -c1-1700
SF_ATTN_FUSED=0 timeout 600 python bench.py --preset MVITv2_S_16x4 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_mvit_unfused.log 2>&1; echo "bench mvit (unfused) rc=$?"; tail -1 gpurun_out/bench_mvit_unfused.log | cut -c1-300
SF_ATTN_FUSED=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v13_mvit -- python bench.py --preset MVITv2_S_16x4 --batch 32 --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof_mvit.log 2>&1; echo "rocprof mvit rc=$?"
