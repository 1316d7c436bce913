#!/bin/bash
# round 2, visit 27: thin forward / data-gradient kernel (independent waves): parity on hardware, microbench, in-step A/B
mkdir -p gpurun_out/v27
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu --tb=short -k "thin or small_shapes or igemm2" > gpurun_out/v27/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/v27/pytest.log | cut -c1-300
run() { timeout 300 env "$@" python tools/microbench.py --iters 5 --no-bn --filter fast > gpurun_out/v27/mb_$TAG.txt 2>&1; echo "== $TAG"; grep -E "^(s[2-5]\.fast)" gpurun_out/v27/mb_$TAG.txt | awk '{printf "%s %s %s fwd %s dgrad ", $1,$2,$3,$9; for(i=1;i<=NF;i++) if($i=="dgrad") printf "%s | ", $(i+1); print ""}' | tr '\n' ' '; echo; }
TAG=off;   run SF_IGEMM2T=0
TAG=on;    run SF_IGEMM2T=1
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-secondary"
brun() { timeout 300 env "$@" $B > gpurun_out/v27/$TAG.json 2>gpurun_out/v27/$TAG.err; echo "$TAG: $(python -c "import json;d=json.loads(open('gpurun_out/v27/$TAG.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"; }
TAG=bench_off; brun SF_IGEMM2T=0
TAG=bench_on;  brun SF_IGEMM2T=1
TAG=bench_on_b1024; brun SF_IGEMM2T_BLOCKS=1024
TAG=bench_on_b768; brun SF_IGEMM2T_BLOCKS=768
