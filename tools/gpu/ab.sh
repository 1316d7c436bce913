#!/bin/bash
# In-step A/B runner (run on the GPU box through gpurun): every variant = a label and a comma-separated environment, all variants
# interleaved `ROUNDS` times through bench.py (cold operands, captured step: what a warm microbenchmark cannot show).
#   tools/gpu/ab.sh OUTDIR [bench.py flags ...] -- "base:X=1" "streams off:SF_PATHWAY_STREAMS=0" ...
#   ROUNDS=3 tools/gpu/ab.sh gpurun_out/x --preset MVITv2_S_16x4 -- "a:..." "b:..."
# Prints `label clips/s ms_per_step final_loss` per run and appends them to OUTDIR/ab.txt.  Tuning knobs of the native library
# (csrc/sf_api.hip: tune_knob) are live only in the diagnostic build: `python -m slowfast_amd.build_ext --diag` and
# SFAMD_LIBRARY=slowfast_amd/libsfamd_diag.so in the variant's environment.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp PYTHONPATH=$PWD
D=$1; shift; mkdir -p "$D"
FLAGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do FLAGS+=("$1"); shift; done
shift
B="--steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-profile"
for R in $(seq 1 ${ROUNDS:-2}); do
  for V in "$@"; do
    L=${V%%:*}; E=${V#*:}
    env ${E//,/ } timeout 300 python bench.py "${FLAGS[@]}" $B 2> "$D/err.txt" | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a "$D/ab.txt"
  done
done
