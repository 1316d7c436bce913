#!/bin/bash
# round 2, GPU visit 3: igemm2 K order (chunk outer, tap inner) and tile / K-step variants, per layer and end to end.
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
F="--filter slow"
run_mb() { # name, env...
  local name=$1; shift
  env "$@" timeout 120 python tools/microbench.py --json gpurun_out/mb3_$name.json --no-bn > gpurun_out/mb3_$name.txt 2>&1; echo "mb $name rc=$? $(tail -1 gpurun_out/mb3_$name.txt)"
}
run_mb default SF_X=0
run_mb bk32 SF_IGEMM2_BK=32
run_mb mink256 SF_IGEMM2_MINK=256
run_mb mink256_bk32 SF_IGEMM2_MINK=256 SF_IGEMM2_BK=32
run_mb all64 SF_IGEMM2_BK64_MINK=0
for V in "default SF_X=0" "bk32 SF_IGEMM2_BK=32" "mink1024 SF_IGEMM2_MINK=1024" "mink256 SF_IGEMM2_MINK=256" "all64 SF_IGEMM2_BK64_MINK=0"; do
  set -- $V
  env $2 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > gpurun_out/bench3_$1.log 2>&1; echo "bench $1 rc=$? $(tail -1 gpurun_out/bench3_$1.log | cut -c1-200)"
done
