#!/bin/bash
# round 3 visit 11: version-2 depthwise stencils (out-of-range taps redirected by address) against version 1, X3D-M and MViTv2-S,
# two rounds each; the kernel + model tests of the GPU suite with version 2 on
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/v11; export TMPDIR=/tmp
V2="SF_DW_FWD_V2=1 SF_DW_DGRAD_V2=1 SF_DW_WGRAD_V2=1"
env $V2 timeout 900 python -m pytest tests/test_tokens_gpu.py tests/test_zy_new_families_gpu.py "tests/test_model_gpu.py::test_x3d_matches_reference" "tests/test_model_gpu.py::test_mvit_matches_reference" -x -q -m gpu > gpurun_out/v11/pytest_gpu.log 2>&1
echo "pytest gpu (dw v2) rc=$?"; tail -2 gpurun_out/v11/pytest_gpu.log
OUT=gpurun_out/v11/dw_v2_ab.txt
: > $OUT
B="python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline"
for R in 1 2; do
  for V in "SF_DUMMY=1" "SF_DW_FWD_V2=1" "SF_DW_DGRAD_V2=1" "SF_DW_WGRAD_V2=1" "$V2"; do
    env $V timeout 200 $B --preset X3D_M --batch 64 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('x3d [$V]', d['value'], d['ms_per_step'])" | tee -a $OUT
    env $V timeout 200 $B --preset MVITv2_S_16x4 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit [$V]', d['value'], d['ms_per_step'])" | tee -a $OUT
  done
done
echo "exit 0"
