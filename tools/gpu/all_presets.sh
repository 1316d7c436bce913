#!/bin/bash
# every BASELINE config as a bench preset on the current code (one JSON line each)
mkdir -p gpurun_out/presets
export PYTHONPATH=$PWD TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
run() { timeout 400 $B "$@" > gpurun_out/presets/$TAG.json 2> gpurun_out/presets/$TAG.err; echo "$TAG: $(python -c "import json;d=json.loads(open('gpurun_out/presets/$TAG.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline']['frac'],d.get('model_roofline',{}).get('frac'))" 2>&1 | tail -1)"; }
TAG=x3d_m_b64;        run --preset X3D_M --batch 64
TAG=c2d_b32;          run --preset C2D_8x8_R50 --batch 32
TAG=r101_nl_ava_b16;  run --preset SLOWFAST_32x2_R101_50_50 --batch 16
TAG=mvit_b_16x4_b32;  run --preset MVIT_B_16x4_CONV --batch 32
