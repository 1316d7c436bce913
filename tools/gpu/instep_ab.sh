#!/bin/bash
# in-step A/B of dispatch / launch knobs through bench.py (cold operands: what a warm microbenchmark cannot show)
mkdir -p gpurun_out/ab
export PYTHONPATH=$PWD TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-secondary"
run() { timeout 300 env "$@" $B $P > gpurun_out/ab/$TAG.json 2>/dev/null; echo "$TAG: $(python -c "import json;d=json.loads(open('gpurun_out/ab/$TAG.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"; }
P=""
TAG=sf_base;      run SF_DUMMY=1
TAG=sf_occ3;      run SF_IGEMM_OCC4=0
P="--preset MVITv2_S_16x4"
TAG=mvit_base;    run SF_DUMMY=1
TAG=mvit_occ3;    run SF_IGEMM_OCC4=0
P="--preset X3D_M --batch 64"
TAG=x3d_base;     run SF_DUMMY=1
