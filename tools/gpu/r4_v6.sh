#!/bin/bash
# round 4 visit 6: A/B of dispatch knobs with no new kernels behind them -- second-generation GEMM for K = 384 / 192 (SF_IGEMM2_MINK),
# 256 x 256 tiles (SF_IGEMM2_BN256: all eligible layers / only grids of >= 512 tiles) -- plus parity of the new relpos kernels.
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v6; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_kernels_gpu.py -k "attention or dw or igemm2 or conv_fwd or conv_dgrad" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $D/pytest.log | cut -c1-300
SF_IGEMM2_BN256=1 timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_kernels_gpu.py tests/test_model_gpu.py -k "igemm2 or conv_fwd or conv_dgrad or slowfast_wc" > $D/pytest_bn256.log 2>&1; echo "pytest bn256 rc=$?"; tail -3 $D/pytest_bn256.log | cut -c1-300
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
run() { # label, env..., -- args
  L=$1; shift
  env "$@" timeout 300 python bench.py $ARGS $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
}
for R in 1 2; do
  ARGS="--preset MVITv2_S_16x4"
  run "mvit base" X=1
  run "mvit mink=384" SF_IGEMM2_MINK=384
  run "mvit mink=192" SF_IGEMM2_MINK=192
  run "mvit bn256=1" SF_IGEMM2_BN256=1
  run "mvit bn256>=512tiles" SF_IGEMM2_BN256=512
  ARGS=""
  run "slowfast base" X=1
  run "slowfast bn256=1" SF_IGEMM2_BN256=1
  run "slowfast bn256>=256tiles" SF_IGEMM2_BN256=256
  run "slowfast bn256>=512tiles" SF_IGEMM2_BN256=512
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/prof_mvit -o p -- python $GRAFT_REPO_ROOT/bench.py --preset MVITv2_S_16x4 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary > $GRAFT_REPO_ROOT/$D/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
F=$(find $D/prof_mvit -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$F" $D/r4_v6_mvit_kernel_stats.md "round 4 visit 6: MViTv2-S bench command (3 timed + 2 warm-up steps), rocprofv3 --kernel-trace --stats" > /dev/null 2>&1
head -48 $D/r4_v6_mvit_kernel_stats.md | tail -41 | cut -c1-150
find $D -name "*.csv" -size +1M -delete
echo "exit 0"
