#!/bin/bash
# MFMA utilisation per SlowFast-R50 layer geometry and entry point: one rocprofv3 --pmc pass per layer of tools/microbench.py
mkdir -p gpurun_out/pmc_layers
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_layers/per_layer_mfma.md
echo "# MFMA utilisation per layer geometry and entry point (SlowFast-8x8-R50, batch 32)" > $OUT
echo "" >> $OUT
echo "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over tools/microbench.py --markers --iters 2 --filter <layer>; MfmaUtil % = busy / (active x 128), time-weighted over the convolution kernels of the phase (tools/pmc_per_layer.py)" >> $OUT
echo "" >> $OUT
echo "| layer | fwd MfmaUtil % | dgrad % | wgrad % | kernels (fwd / dgrad / wgrad) |" >> $OUT
echo "|---|---:|---:|---:|---|" >> $OUT
cd /tmp
python - <<'PY' > /tmp/layers.txt
import sys
sys.path.insert(0, "/root/repo/tools")
import microbench
for l in microbench.LAYERS: print(l[0])
PY
i=0
while read -r L; do
  i=$((i+1))
  rm -rf /tmp/pl; timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pl -o p -- python $R/tools/microbench.py --iters 2 --no-bn --markers --filter "$L" > /dev/null 2>&1
  F=$(find /tmp/pl -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python $R/tools/pmc_per_layer.py "$F" "$L" >> $OUT
done < /tmp/layers.txt
cat $OUT | cut -c1-200
