#!/bin/bash
# GPU visit 12: fused attention (sf_attn_fwd / sf_attn_bwd, ABI v9): parity + MViT A/B against the unfused chain.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -12 | cut -c1-600
timeout 600 python bench.py --preset MVITv2_S_16x4 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_mvit.log 2>&1; echo "bench mvit rc=$?"; tail -1 gpurun_out/bench_mvit.log | cut -c1-2200
SF_ATTN_FUSED=0 timeout 600 python bench.py --preset MVITv2_S_16x4 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_mvit_unfused.log 2>&1; echo "bench mvit (unfused) rc=$?"; tail -1 gpurun_out/bench_mvit_unfused.log | cut -c1-400
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1v12_mvit -- python bench.py --preset MVITv2_S_16x4 --batch 32 --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/rocprof_mvit.log 2>&1; echo "rocprof mvit rc=$?"
