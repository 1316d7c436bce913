#!/bin/bash
# kernel trace + PMC passes of the depthwise kernels through tools/dw_bench.py (run on the GPU box through gpurun):
#   tools/gpu/pmc_dw.sh OUTDIR [dw_bench flags ...]      e.g.  tools/gpu/pmc_dw.sh gpurun_out/pmc_dw --only s3
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD; D=$R/$1; shift; mkdir -p "$D"
cd /tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o p -- python $R/tools/dw_bench.py --iters 6 "$@" > $D/bench.txt 2>&1
P1="GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
timeout 250 rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $D/pmc1 -o p -- python $R/tools/dw_bench.py --iters 2 "$@" > /dev/null 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $D/pmc2 -o p -- python $R/tools/dw_bench.py --iters 2 "$@" > /dev/null 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/pmc3 -o p -- python $R/tools/dw_bench.py --iters 2 "$@" > /dev/null 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/pmc4 -o p -- python $R/tools/dw_bench.py --iters 2 "$@" > /dev/null 2>&1
cd $R
F=$(find $D/trace -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" $D/stats.md "dw_bench $*" > /dev/null 2>&1
grep -E "dw" $D/stats.md | cut -c1-170
for P in pmc1 pmc2 pmc3 pmc4; do
  FM=$(find $D/$P -name "*counter_collection.csv" | head -1)
  python tools/pmc_metric.py $D/$P.md "dw_bench $* $P" "$FM" > /dev/null 2>&1
  grep -E "dw" $D/$P.md | cut -c1-260
done
find $D -name "*.csv" -size +1M -delete; find $D -name "*.db" -delete
