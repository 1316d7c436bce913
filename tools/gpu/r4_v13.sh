#!/bin/bash
# round 4 visit 13: the mvit_tiny AdamW train-step GPU test went non-finite in the full-suite run -- find the switch
D=gpurun_out/v13; mkdir -p $D
run() { echo "== $*"; env "$@" timeout 300 python tools/gpu/r4_nan_probe.py $MODE 5 2>&1 | grep -E "^(graph|eager) it|Error|error" | cut -c1-400; }
(
MODE=graph; run PROBE_DIRTY=1
MODE=eager; run PROBE_DIRTY=1
MODE=graph; run PROBE_DIRTY=0
MODE=graph; run SF_LN_BIAS_SUMS=0
MODE=graph; run SF_LN_RU=1
MODE=graph; run SF_FIN_BATCH=0
MODE=graph; run SF_MVIT_RESID32=0
MODE=graph; run SF_ATTN_DKV_KT=1
MODE=graph; run SF_DW_TILED=0
) > $D/probe.txt 2>&1
cat $D/probe.txt
timeout 600 python -m pytest -q -m gpu -x --tb=short tests/test_step.py > $D/pytest_step.log 2>&1; echo "pytest step rc=$?"; tail -5 $D/pytest_step.log | cut -c1-300
