#!/bin/bash
# round 2, visit 25: thin weight gradient: deeper rings again, now that the scalar chain is gone
mkdir -p gpurun_out/v25
export PYTHONPATH=$PWD TMPDIR=/tmp
run() { timeout 300 env "$@" python tools/microbench.py --iters 5 --no-bn --filter fast > gpurun_out/v25/mb_$TAG.txt 2>&1; echo "== $TAG"; grep -E "^(s[2-5]\.fast)" gpurun_out/v25/mb_$TAG.txt | awk '{for(i=1;i<=NF;i++) if($i=="wgrad") printf "%s %s %s %s | ", $1,$2,$3,$(i+1); print ""}' | tr '\n' ' '; echo; }
TAG=b512;        run SF_WGRAD2T_BLOCKS=512
TAG=deep256;     run SF_WGRAD2T_DEEP=1 SF_WGRAD2T_BLOCKS=256
TAG=deep512;     run SF_WGRAD2T_DEEP=1 SF_WGRAD2T_BLOCKS=512
