#!/bin/bash
# in-step A/B of dispatch knobs through bench.py (cold operands: what a warm microbenchmark cannot show; visit 26)
mkdir -p gpurun_out/ab
export PYTHONPATH=$PWD TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-secondary"
run() { timeout 300 env "$@" $B > gpurun_out/ab/$TAG.json 2>/dev/null; echo "$TAG: $(python -c "import json;d=json.loads(open('gpurun_out/ab/$TAG.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"; }
TAG=base;            run SF_DUMMY=1
TAG=base2;           run SF_DUMMY=1
TAG=i2_mink256;      run SF_IGEMM2_MINK=256
TAG=i2_mink128;      run SF_IGEMM2_MINK=128
TAG=i2_bk32;         run SF_IGEMM2_BK=32
TAG=i2_bk64;         run SF_IGEMM2_BK=64
TAG=w2_mink64;       run SF_WGRAD2_MINK=64
TAG=w2_mink128;      run SF_WGRAD2_MINK=128
TAG=w2_b768;         run SF_WGRAD2_BLOCKS=768
TAG=w2_b1536;        run SF_WGRAD2_BLOCKS=1536
TAG=w2t_b1024;       run SF_WGRAD2T_BLOCKS=1024
TAG=w2t_norr;        run SF_WGRAD2T_RR=0
TAG=mvit_base;       run SF_DUMMY=1 SF_X=1; 
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-secondary --preset MVITv2_S_16x4"
TAG=mvit;            run SF_DUMMY=1
TAG=mvit_i2_mink256; run SF_IGEMM2_MINK=256
TAG=mvit_i2_mink96;  run SF_IGEMM2_MINK=96
TAG=mvit_w2_mink96;  run SF_WGRAD2_MINK=96
