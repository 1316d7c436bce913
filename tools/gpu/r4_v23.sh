#!/bin/bash
# round 4 visit 23: what bounds the new token kernels (tiled depthwise weight gradient, LayerNorm backward)?  token_bench.py at
# the MViTv2-S stage-3 pooling shapes and the LayerNorm shapes under rocprofv3 --pmc (one counter group per pass)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v23; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
cd /tmp
R=$GRAFT_REPO_ROOT
i=0
for G in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/$D/pmc$i -o p -- python $R/tools/token_bench.py --iters 2 --only stage3 > /dev/null 2>&1; echo "pmc$i rc=$?"
  FM=$(find $R/$D/pmc$i -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_metric.py $R/$D/pmc$i.md "round 4 visit 23, token_bench.py --iters 2 --only stage3 (MViTv2-S stage-3 pooling shapes, batch 32), pass $i" "$FM" > /dev/null 2>&1
  grep -E "dwtile|dwconv|kernel \||---" $R/$D/pmc$i.md | cut -c1-260
done
cd $R; find $D -name "*.csv" -size +1M -delete
