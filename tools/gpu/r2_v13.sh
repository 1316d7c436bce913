#!/bin/bash
# round 2, GPU visit 13: igemm2 K-loop variants (fragment reads before the copies / s_setprio), wgrad2 split rule.
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
run_mb() { local name=$1; shift; env "$@" timeout 120 python tools/microbench.py --json gpurun_out/mb13_$name.json --no-bn > gpurun_out/mb13_$name.txt 2>&1; echo "mb $name rc=$? $(tail -1 gpurun_out/mb13_$name.txt)"; }
run_mb v0 SF_IGEMM2_VARIANT=0
run_mb v1 SF_IGEMM2_VARIANT=1
run_mb v2 SF_IGEMM2_VARIANT=2
run_mb v3 SF_IGEMM2_VARIANT=3
for V in "v0 SF_IGEMM2_VARIANT=0" "v1 SF_IGEMM2_VARIANT=1" "v3 SF_IGEMM2_VARIANT=3"; do
  set -- $V
  env $2 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-secondary > gpurun_out/bench13_$1.log 2>&1; echo "bench $1 rc=$? $(tail -1 gpurun_out/bench13_$1.log | cut -c1-200)"
done
