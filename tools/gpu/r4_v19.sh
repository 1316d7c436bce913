#!/bin/bash
# round 4 visit 19: tiled depthwise weight gradient with 32-float x positions and evenly cut row segments (all MViTv2-S shapes)
D=gpurun_out/v19; mkdir -p $D
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_tokens_gpu.py tests/test_zz_dwconv_shapes_gpu.py tests/test_model_gpu.py -k "dw or depthwise or mvit_matches or x3d" > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $D/pytest.log | cut -c1-300
for V in 0 1; do
  echo "== SF_DW_WGRAD_TILED=$V" | tee -a $D/r4_v19_dw_bench.txt
  SF_TRACE=$V SF_DW_WGRAD_TILED=$V timeout 200 python tools/token_bench.py --iters 20 --only dw 2>&1 | grep -E "^dwconv|dwtile wgrad" | sort | uniq | tee -a $D/r4_v19_dw_bench.txt
done
for V in 0 1 0 1; do
  SF_DW_WGRAD_TILED=$V timeout 300 python bench.py --preset MVITv2_S_16x4 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('mvit SF_DW_WGRAD_TILED=$V', d['value'], d['ms_per_step'])" | tee -a $D/r4_v19_dw_ab.txt
done
