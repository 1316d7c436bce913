#!/bin/bash
# round 3 visit 12: the stem's LDS-patch direct convolution on the Fast pathway's 8-channel 1x3x3 layer (forward only, experiment)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/v12; export TMPDIR=/tmp
SF_STEM_THIN3=1 timeout 300 python -c "
import torch
from tests import kernel_checks as kc
d=torch.device('cuda:0')
kc.check_conv_fwd(d,(4,8,16,56,56),8,(1,3,3),(1,1,1),(0,1,1))
kc.check_conv_fwd(d,(2,8,5,30,30),16,(1,3,3),(1,1,1),(0,1,1))
kc.check_conv_fwd(d,(2,8,6,14,14),8,(3,3,3),(1,1,1),(1,1,1))
print('thin3 fwd ok')"
for V in 0 1; do
  echo "== SF_STEM_THIN3=$V"
  SF_STEM_THIN3=$V timeout 300 python tools/microbench.py --no-bn --iters 8 --filter "fast b" 2>&1 | grep -E "^s[2-5]" | cut -c1-150
done
echo "exit 0"
