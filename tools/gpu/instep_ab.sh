#!/bin/bash
# in-step A/B of dispatch / launch knobs through bench.py (cold operands: what a warm microbenchmark cannot show)
mkdir -p gpurun_out/ab
export PYTHONPATH=$PWD TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-secondary"
run() { timeout 300 env "$@" $B > gpurun_out/ab/$TAG.json 2>/dev/null; echo "$TAG: $(python -c "import json;d=json.loads(open('gpurun_out/ab/$TAG.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"; }
TAG=base;            run SF_DUMMY=1
TAG=bnblocks512;     run SF_BN_BWD_BLOCKS=512
TAG=bnblocks2048;    run SF_BN_BWD_BLOCKS=2048
TAG=fold256;         run SF_FOLD_ABOVE=256
TAG=fold16384;       run SF_FOLD_ABOVE=16384
TAG=wgrad_stream;    run SF_WGRAD_STREAM=1
TAG=w2t_minrows;     run SF_WGRAD2T_MINROWS=4096
TAG=wgrad_blocks512; run SF_WGRAD_BLOCKS=512
TAG=wgrad_blocks2048; run SF_WGRAD_BLOCKS=2048
TAG=base2;           run SF_DUMMY=1
