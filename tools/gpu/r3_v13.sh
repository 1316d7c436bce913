#!/bin/bash
# round 3 visit 13: the LDS-patch direct convolution (sf_stem.h) on the Fast pathway's 8-channel 1x3x3 layer, all three directions
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/v13; export TMPDIR=/tmp
SF_STEM_THIN3=1 timeout 300 python -c "
import torch
from tests import kernel_checks as kc
d=torch.device('cuda:0')
for shp,co,k,p in (((4,8,16,56,56),8,(1,3,3),(0,1,1)), ((2,8,5,30,30),16,(1,3,3),(0,1,1)), ((2,8,6,14,14),8,(3,3,3),(1,1,1))):
    kc.check_conv_fwd(d,shp,co,k,(1,1,1),p); kc.check_conv_wgrad(d,shp,co,k,(1,1,1),p)
for shp,co,k,p in (((4,8,16,56,56),8,(1,3,3),(0,1,1)), ((2,16,5,30,30),8,(1,3,3),(0,1,1)), ((2,8,6,14,14),8,(3,3,3),(1,1,1))):
    kc.check_conv_dgrad(d,shp,co,k,(1,1,1),p); kc.check_conv_dgrad_bn(d,shp,co,k,p)
print('thin3 ok')"
OUT=gpurun_out/v13/thin3_ab.txt
: > $OUT
for V in 0 1; do
  echo "== SF_STEM_THIN3=$V" | tee -a $OUT
  SF_STEM_THIN3=$V timeout 300 python tools/microbench.py --no-bn --iters 8 --filter "fast b" 2>&1 | grep -E "^s[2-5]" | cut -c1-170 | tee -a $OUT
done
SF_STEM_THIN3=1 timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "blocks_strict or slowfast_wc or SLOWFAST_8x8" 2>&1 | tail -2
B="python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
for R in 1 2; do
  for V in 0 1; do
    SF_STEM_THIN3=$V timeout 200 $B 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('slowfast thin3=$V', d['value'], d['ms_per_step'])" | tee -a $OUT
  done
done
echo "exit 0"
