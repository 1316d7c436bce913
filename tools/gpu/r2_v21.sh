#!/bin/bash
# round 2, visit 21: thin weight gradient: round-robin stage assignment, workgroup count sweep (Fast-pathway layers only)
mkdir -p gpurun_out/v21
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu --tb=short -k "wgrad2_thin or small_shapes" > gpurun_out/v21/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/v21/pytest.log | cut -c1-200
run() { timeout 300 env "$@" python tools/microbench.py --iters 5 --no-bn --filter fast > gpurun_out/v21/mb_$TAG.txt 2>&1; echo "== $TAG"; grep -E "^(s[2-5]\.fast)" gpurun_out/v21/mb_$TAG.txt | awk '{for(i=1;i<=NF;i++) if($i=="wgrad") printf "%s %s %s %s | ", $1,$2,$3,$(i+1); print ""}' | tr '\n' ' '; echo; }
TAG=old;        run SF_WGRAD2T=0
TAG=rr1024;     run SF_WGRAD2T_BLOCKS=1024
TAG=norr1024;   run SF_WGRAD2T_RR=0 SF_WGRAD2T_BLOCKS=1024
TAG=rr512;      run SF_WGRAD2T_BLOCKS=512
TAG=rr768;      run SF_WGRAD2T_BLOCKS=768
TAG=rr2048;     run SF_WGRAD2T_BLOCKS=2048
TAG=rr4096;     run SF_WGRAD2T_BLOCKS=4096
