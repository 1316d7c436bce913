#!/bin/bash
# GPU visit 16: A/B of split-K block target (SF_WGRAD_BLOCKS) and BatchNorm-backward reduce blocks (SF_BN_BWD_BLOCKS).
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
run() { # name preset batch env...
  local name=$1 preset=$2 batch=$3; shift 3
  env "$@" timeout 600 python bench.py --preset $preset --batch $batch --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-profile > gpurun_out/ab_$name.log 2>&1
  echo "$name rc=$? $(grep -h '^{' gpurun_out/ab_$name.log | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])' 2>/dev/null)"
}
run sf_base SLOWFAST_8x8_R50 32 SF_X=0
run sf_wg512 SLOWFAST_8x8_R50 32 SF_WGRAD_BLOCKS=512
run sf_wg2048 SLOWFAST_8x8_R50 32 SF_WGRAD_BLOCKS=2048
run sf_bn2048 SLOWFAST_8x8_R50 32 SF_BN_BWD_BLOCKS=2048
run sf_bn4096 SLOWFAST_8x8_R50 32 SF_BN_BWD_BLOCKS=4096
run mvit_base MVITv2_S_16x4 32 SF_X=0
run mvit_wg512 MVITv2_S_16x4 32 SF_WGRAD_BLOCKS=512
run mvit_wg2048 MVITv2_S_16x4 32 SF_WGRAD_BLOCKS=2048
run x3d_base X3D_M 64 SF_X=0
run x3d_bn4096 X3D_M 64 SF_BN_BWD_BLOCKS=4096
run x3d_wg512 X3D_M 64 SF_WGRAD_BLOCKS=512
