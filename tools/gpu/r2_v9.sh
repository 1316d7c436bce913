#!/bin/bash
# round 2, GPU visit 9: yardstick-free parity -- well-conditioned golden cases and BASELINE configs 2-5 at full size (batch 2).
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
rm -f gpurun_out/parity_wc.jsonl
SF_PARITY_REPORT=$PWD/gpurun_out/parity_wc.jsonl timeout 900 python -m pytest tests/test_model_gpu.py -q --tb=short -s -k "well_conditioned or full_size_batch2" > gpurun_out/pytest9.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error|logits_l2" gpurun_out/pytest9.log | tail -14 | cut -c1-420
