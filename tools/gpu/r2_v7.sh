#!/bin/bash
# round 2, GPU visit 7: PMC counters of the new GEMM kernels on single layers (is the K loop L2-, HBM-, LDS- or issue-bound?).
# Counters in their own runs (no trace domains besides --kernel-trace), per MI355X_MICROARCH.md.
mkdir -p gpurun_out/pmc7
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
CMD="python $R/tools/microbench.py --no-bn --iters 2 --filter s4.slow"
pass() { # name, counters...
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc7/$name -o p -- $CMD > $R/gpurun_out/pmc7/$name.log 2>&1; echo "pass $name rc=$?"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
cd $R
for n in fetch write tcc sq1 lds tcp; do
  F=$(find gpurun_out/pmc7/$n -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python tools/pmc_metric.py gpurun_out/pmc7/$n.md "visit 7 PMC pass $n: microbench s4.slow layers" "$F" > /dev/null 2>&1
  head -14 gpurun_out/pmc7/$n.md | cut -c1-260
done
find gpurun_out/pmc7 -name "*.csv" -size +2M -delete
