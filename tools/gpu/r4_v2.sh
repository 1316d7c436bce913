#!/bin/bash
# round 4 visit 2: key-side attention backward with two key tiles per wave (SF_ATTN_DKV_KT) and batched column-sum finalizes
# (SF_FIN_BATCH): GPU parity of the touched paths, A/B on the MViTv2-S step, rocprofv3 kernel stats of the new default.
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v2; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_model_gpu.py -k "attention or mvit or MVIT or rows32" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $D/pytest.log | cut -c1-300
SF_ATTN_DKV_KT=1 timeout 600 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py -k "attention" > $D/pytest_kt1.log 2>&1; echo "pytest kt1 rc=$?"; tail -1 $D/pytest_kt1.log
B="python bench.py --preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/mvit_ab.txt
for R in 1 2; do for V in "1 0" "2 0" "1 1" "2 1"; do set -- $V
  SF_ATTN_DKV_KT=$1 SF_FIN_BATCH=$2 timeout 300 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit dkv_kt=$1 fin_batch=$2', d['value'], d['ms_per_step'])" | tee -a $D/mvit_ab.txt
done; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/prof_mvit -o p -- python $GRAFT_REPO_ROOT/bench.py --preset MVITv2_S_16x4 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary > $GRAFT_REPO_ROOT/$D/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
F=$(find $D/prof_mvit -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" $D/r4_v2_mvit_kernel_stats.md "round 4 visit 2: MViTv2-S bench command (3 timed + 2 warm-up steps), KT=2 key-side attention backward + batched finalizes, rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
head -40 $D/r4_v2_mvit_kernel_stats.md | cut -c1-150
find $D -name "*.csv" -size +1M -delete
echo "exit 0"
