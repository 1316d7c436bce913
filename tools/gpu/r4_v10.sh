#!/bin/bash
# round 4 visit 10: forward / dQ attention kernels with direct-to-LDS double-buffered K / V / OH chunks (one barrier per chunk)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v10; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_model_gpu.py -k "attention or mvit_matches or mvit_v1 or MVIT" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
timeout 120 python tools/token_bench.py --iters 20 --only attn 2>&1 | grep "^attn" | tee $D/attn.txt
B="--preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
for R in 1 2; do
  timeout 300 python bench.py $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/prof_mvit -o p -- python $GRAFT_REPO_ROOT/bench.py --preset MVITv2_S_16x4 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary > $GRAFT_REPO_ROOT/$D/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
F=$(find $D/prof_mvit -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" $D/r4_v10_mvit_kernel_stats.md "round 4 visit 10: MViTv2-S bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
head -30 $D/r4_v10_mvit_kernel_stats.md | tail -23 | cut -c1-150
find $D -name "*.csv" -size +1M -delete
echo "exit 0"
