#!/bin/bash
# round 4 visit 17: finalize kernels with eight table rows in flight + shuffle fold (one barrier) against the previous binary
# (tools/gpu/ab/libsfamd_old.so = the library of the commit before), same box; LayerNorm forward block-count sweep
D=gpurun_out/v17; mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_tokens_gpu.py tests/test_model_gpu.py -k "bn or batchnorm or norm or finalize or colsum or slowfast_wc or c2d_wc or mvit_matches or x3d" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
OLD="SFAMD_LIBRARY=$R/tools/gpu/ab/libsfamd_old.so SF_ALLOW_STALE_LIBRARY=1"
for P in SLOWFAST_8x8_R50 MVITv2_S_16x4 X3D_M; do
  for V in old new old new; do
    E="SF_NOOP=1"; [ $V = old ] && E="$OLD"
    env $E timeout 300 python bench.py --preset $P --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$P finalize $V', d['value'], d['ms_per_step'])" | tee -a $D/r4_v17_finalize_ab.txt
  done
done
cd /tmp
for V in old new; do
  E="SF_NOOP=1"; [ $V = old ] && E="$OLD"
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/prof_$V -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary --preset SLOWFAST_8x8_R50 --batch 32 > $R/$D/rocprof_$V.log 2>&1; echo "rocprof $V rc=$?"
done
cd $R
for V in old new; do
  F=$(find $D/prof_$V -name "*kernel_stats.csv" | head -1)
  python tools/rocprof_summary.py "$F" $D/r4_v17_slowfast_kernel_stats_$V.md "round 4 visit 17: SlowFast bench command (3 timed + 2 warm-up steps), finalize kernels $V" > /dev/null 2>&1
  echo "== $V"; grep -E "finalize|part_fold|total kernel" $D/r4_v17_slowfast_kernel_stats_$V.md | cut -c1-140
done
for B in 2048 4096 8192; do
  echo "== SF_LN_FWD_BLOCKS=$B" | tee -a $D/r4_v17_ln_fwd_blocks.txt
  SF_LN_FWD_BLOCKS=$B timeout 200 python tools/token_bench.py --iters 20 --only ln 2>&1 | grep layernorm | cut -c1-60 | tee -a $D/r4_v17_ln_fwd_blocks.txt
done
