#!/bin/bash
# round 3 visit 9: per-shape timings of the MViTv2-S Linear layers (the round-1 GEMM kernel on K = 96..384)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/v9; export TMPDIR=/tmp
timeout 600 python tools/gemm_bench.py --md gpurun_out/v9/r3_v9_mvit_gemm_shapes.md 2>&1 | tail -25
for MK in 64 96; do echo "== SF_IGEMM2_MINK=$MK"; SF_IGEMM2_MINK=$MK timeout 600 python tools/gemm_bench.py 2>&1 | grep -E "s1 |s2 |weighted" ; done
echo "exit 0"
