#!/bin/bash
# round 3 visit 6: which part of sf_igemm2_kernel bounds its K loop?  Diagnostic ablations (SF_IGEMM2_ABLATE bits: 1 no copies in
# the loop, 2 no LDS reads / MFMAs, 4 LDS reads but no MFMAs, 8 no epilogue, 16 no barrier); results of ablated runs are garbage.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/v6; export TMPDIR=/tmp
OUT=gpurun_out/v6/igemm2_ablation.txt
: > $OUT
for A in 0 1 2 4 8 16 3 11 17 24; do
  echo "== SF_IGEMM2_ABLATE=$A" | tee -a $OUT
  SF_IGEMM2_ABLATE=$A timeout 300 python tools/microbench.py --no-bn --iters 6 --filter "slow b|slow a" 2>&1 | grep -E "^s[2-5]" | cut -c1-125 | tee -a $OUT
done
echo "exit 0"
