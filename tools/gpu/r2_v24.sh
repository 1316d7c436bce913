#!/bin/bash
# round 2, visit 24: thin weight gradient after the scalar-load fix: workgroup counts, M0 left in place; kernel tests
mkdir -p gpurun_out/v24
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu --tb=short -k "igemm2 or wgrad" > gpurun_out/v24/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/v24/pytest.log | cut -c1-200
run() { timeout 300 env "$@" python tools/microbench.py --iters 5 --no-bn --filter fast > gpurun_out/v24/mb_$TAG.txt 2>&1; echo "== $TAG"; grep -E "^(s[2-5]\.fast)" gpurun_out/v24/mb_$TAG.txt | awk '{for(i=1;i<=NF;i++) if($i=="wgrad") printf "%s %s %s %s | ", $1,$2,$3,$(i+1); print ""}' | tr '\n' ' '; echo; }
TAG=old;        run SF_WGRAD2T=0
TAG=b512;       run SF_WGRAD2T_BLOCKS=512
TAG=b768;       run SF_WGRAD2T_BLOCKS=768
TAG=b1024;      run SF_WGRAD2T_BLOCKS=1024
TAG=b1536;      run SF_WGRAD2T_BLOCKS=1536
TAG=b2048;      run SF_WGRAD2T_BLOCKS=2048
TAG=norr1024;   run SF_WGRAD2T_RR=0 SF_WGRAD2T_BLOCKS=1024
timeout 600 python tools/microbench.py --iters 5 --no-bn --md gpurun_out/v24/r2_v24_per_geometry.md > gpurun_out/v24/microbench.txt 2>&1; grep "weighted totals" gpurun_out/v24/r2_v24_per_geometry.md
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-secondary"
timeout 300 $B > gpurun_out/v24/bench.json 2>/dev/null; echo "bench: $(python -c "import json;d=json.loads(open('gpurun_out/v24/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
timeout 300 $B --preset MVITv2_S_16x4 > gpurun_out/v24/bench_mvit.json 2>/dev/null; echo "mvit: $(python -c "import json;d=json.loads(open('gpurun_out/v24/bench_mvit.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
