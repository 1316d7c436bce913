#!/bin/bash
# GPU visit 15: RoI head (ABI v10), gradient clipping in the MViT step, thread-local capture under RCCL, MFMA/VALU PMC.
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short > gpurun_out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error|FAILED" gpurun_out/pytest_gpu.log | tail -12 | cut -c1-600
timeout 900 python bench.py --preset SLOWFAST_32x2_R101_50_50 --batch 16 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_ava.log 2>&1; echo "bench ava rc=$?"; tail -1 gpurun_out/bench_ava.log | cut -c1-900
timeout 600 python bench.py --preset MVITv2_S_16x4 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench_mvit.log 2>&1; echo "bench mvit rc=$?"; tail -1 gpurun_out/bench_mvit.log | cut -c1-300
rm -rf gpurun_out/pmc2
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc2 -o sf -- python bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-profile > gpurun_out/pmc_mfma_sf.log 2>&1; echo "pmc mfma slowfast rc=$?"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc2 -o mvit -- python bench.py --preset MVITv2_S_16x4 --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-profile > gpurun_out/pmc_mfma_mvit.log 2>&1; echo "pmc mfma mvit rc=$?"
ls gpurun_out/pmc2 | head; tail -3 gpurun_out/pmc_mfma_sf.log | cut -c1-300
