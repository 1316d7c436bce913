#!/bin/bash
# round 4 visit 4: XCD-contiguous block order of the depthwise stencils (SF_DW_XCD) and of the stem weight gradient (SF_STEM_XCD),
# attention LDS pitch D + 16: kernel parity, warm microbenchmark, in-step A/B on MViTv2-S, X3D-M and SlowFast.
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v4; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_zz_dwconv_shapes_gpu.py tests/test_kernels_gpu.py -k "attention or dw or depthwise or stem or x3d" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $D/pytest.log | cut -c1-300
for V in 0 1; do SF_DW_XCD=$V timeout 300 python tools/token_bench.py --iters 10 2>&1 | grep -v amdgpu.ids | sed "s/^/dw_xcd=$V /" | tee -a $D/token_bench.txt; done
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
for R in 1 2; do
  for V in 0 1; do
    SF_DW_XCD=$V timeout 300 python bench.py --preset MVITv2_S_16x4 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit dw_xcd=$V', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
  done
  for V in 0 1; do
    SF_STEM_XCD=$V timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-profile 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('slowfast stem_xcd=$V', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
  done
done
for V in 0 1; do
  SF_DW_XCD=$V timeout 300 python bench.py --preset X3D_M --batch 64 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('x3d dw_xcd=$V', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/prof_mvit -o p -- python $GRAFT_REPO_ROOT/bench.py --preset MVITv2_S_16x4 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary > $GRAFT_REPO_ROOT/$D/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
F=$(find $D/prof_mvit -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" $D/r4_v4_mvit_kernel_stats.md "round 4 visit 4: MViTv2-S bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
head -36 $D/r4_v4_mvit_kernel_stats.md | tail -29 | cut -c1-150
find $D -name "*.csv" -size +1M -delete
echo "exit 0"
