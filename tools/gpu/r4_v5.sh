#!/bin/bash
# round 4 visit 5: LDS-tiled plane-sweep depthwise convolution (sf_dwtile.h; forward + data gradient) against the W-blocked stencils
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v5; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
SF_DW_TILED=2 timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_zz_dwconv_shapes_gpu.py -k "dw or depthwise" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $D/pytest.log | cut -c1-300
for V in 0 1 2; do SF_DW_TILED=$V SF_TRACE=$V timeout 300 python tools/token_bench.py --iters 10 --only dw 2>&1 | grep -v amdgpu.ids | sed "s/^/tiled=$V /" | sort -u | tee -a $D/token_bench.txt; done
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
: > $D/ab.txt
for R in 1 2; do for V in 0 1; do
  SF_DW_TILED=$V timeout 300 python bench.py --preset MVITv2_S_16x4 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit dw_tiled=$V', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
done; done
timeout 600 python -m pytest -q -m gpu -x --tb=short tests/test_model_gpu.py -k "mvit_matches or MVIT" > $D/pytest_model.log 2>&1; echo "pytest model rc=$?"; tail -2 $D/pytest_model.log | cut -c1-200
echo "exit 0"
