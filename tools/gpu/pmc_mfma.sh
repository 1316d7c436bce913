#!/bin/bash
# MFMA-busy / wave-state counters of the bench command of the two headline models (run on the GPU box through gpurun):
#   ROUND=6 bash tools/gpu/pmc_mfma.sh [tag]  ->  gpurun_out/r6_pmc_mfma/r6_<tag>_pmc_mfma_{slowfast,mvit}.md
cd "$GRAFT_REPO_ROOT"; TAG=${1:-final}; D=gpurun_out/r${ROUND:-6}_pmc_mfma; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary"
for P in "SLOWFAST_8x8_R50 32 slowfast" "MVITv2_S_16x4 32 mvit"; do
  set -- $P
  timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/$D/pmc_$3 -o p -- $BENCH --preset $1 --batch $2 > $R/$D/pmc_$3.log 2>&1; echo "pmc mfma $3 rc=$?"
done
cd $R
for n in slowfast mvit; do
  F=$(find $D/pmc_$n -name "*counter_collection.csv" | head -1)
  python tools/pmc_metric.py $D/r${ROUND:-6}_${TAG}_pmc_mfma_$n.md "round ${ROUND:-6} (HEAD): MFMA / wave-state counters, $n bench command" "$F" > /dev/null 2>&1
  grep -E "igemm2_kernel<256, 128|wgrad2_kernel<128|attn_|igemm_kernel<128, 64, 64, true, true, true, false" $D/r${ROUND:-6}_${TAG}_pmc_mfma_$n.md | cut -c1-200
  rm -rf $D/pmc_$n
done
