#!/bin/bash
# round 2, GPU visit 10: re-run the yardstick-free parity tests (slowfast_wc at full width, MViT head in fp32), optimizer tests.
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
rm -f gpurun_out/parity_wc.jsonl
SF_PARITY_REPORT=$PWD/gpurun_out/parity_wc.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q --tb=short -s -k "well_conditioned or full_size_batch2 or mvit_matches" > gpurun_out/pytest10.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error|logits_l2" gpurun_out/pytest10.log | tail -14 | cut -c1-420
timeout 300 python -m pytest tests/test_step.py -q --tb=short > gpurun_out/pytest10b.log 2>&1; echo "pytest step rc=$?"; tail -3 gpurun_out/pytest10b.log
