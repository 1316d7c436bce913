#!/bin/bash
# round 5 visit 46: MFMA / wave-state counters of the MViTv2-S and SlowFast bench commands at HEAD (the attention kernels after their
# loop fixes; round 4's table: profiles/r4/r4_final_pmc_mfma_*.md), plus a second default bench line on another box
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v46; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary"
for P in "MVITv2_S_16x4 mvit" "SLOWFAST_8x8_R50 slowfast"; do
  set -- $P
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/$D/pmc_$2 -o p -- $BENCH --preset $1 --batch 32 > $R/$D/pmc_$2.log 2>&1; echo "pmc $2 rc=$?"
done
cd $R
for n in mvit slowfast; do
  F=$(find $D/pmc_$n -name "*counter_collection.csv" | head -1)
  python tools/pmc_metric.py $D/r5_final_pmc_mfma_$n.md "round 5 (HEAD): MFMA / wave-state counters, $n bench command" "$F" > /dev/null 2>&1
  grep -E "attn|igemm|wgrad2" $D/r5_final_pmc_mfma_$n.md | cut -c1-200 | head -12
  rm -rf $D/pmc_$n
done
timeout 900 python bench.py --no-cpu-baseline > $D/bench2.log 2> $D/bench2.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$D/bench2.log').read().strip().split('\n')[-1]); print('value', d['value'], d['ms_per_step'], 'secondary', d['secondary']['value'], d['secondary']['ms_per_step'])"
echo "exit 0"
