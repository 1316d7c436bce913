#!/bin/bash
# kernel trace + PMC counter passes of a few layers through tools/microbench.py (here: the s2.fast weight gradients, thin vs first kernel; visit 22)
mkdir -p gpurun_out/pmc
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for V in thin old; do
  [ $V = old ] && export SF_WGRAD2T=0 || unset SF_WGRAD2T
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc/trace_$V -o p -- python $R/tools/microbench.py --iters 5 --no-bn --filter s2.fast > $R/gpurun_out/pmc/mb_$V.txt 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc/pmc1_$V -o p -- python $R/tools/microbench.py --iters 2 --no-bn --filter s2.fast > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc/pmc2_$V -o p -- python $R/tools/microbench.py --iters 2 --no-bn --filter s2.fast > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/pmc3_$V -o p -- python $R/tools/microbench.py --iters 2 --no-bn --filter s2.fast > /dev/null 2>&1
done
cd $R
for V in thin old; do
  F=$(find gpurun_out/pmc/trace_$V -name "*kernel_stats.csv" | head -1)
  python tools/rocprof_summary.py "$F" gpurun_out/pmc/stats_$V.md "visit 22 $V" > /dev/null 2>&1
  echo "== $V"; grep -E "wgrad|reduce" gpurun_out/pmc/stats_$V.md | cut -c1-150
  for P in pmc1 pmc2 pmc3; do
    FM=$(find gpurun_out/pmc/${P}_$V -name "*counter_collection.csv" | head -1)
    python tools/pmc_metric.py gpurun_out/pmc/${P}_$V.md "visit 22 $P $V" "$FM" > /dev/null 2>&1
    grep -E "wgrad" gpurun_out/pmc/${P}_$V.md | cut -c1-220
  done
done
find gpurun_out/pmc -name "*.csv" -size +1M -delete
