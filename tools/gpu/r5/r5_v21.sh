#!/bin/bash
# round 5 visit 21: HBM-cold timing of the stage-3 token GEMMs under the three GEMM generations
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v21; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
for V in "default:X=1" "igemm2 (mink 384):SF_IGEMM2_MINK=384" "igemm3 (mink 384):SF_IGEMM2_MINK=384,SF_IGEMM3=1"; do
  L=${V%%:*}; E=${V#*:}
  echo "== $L" | tee -a $D/cold.txt
  env ${E//,/ } timeout 300 python tools/gemm_cold_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $D/cold.txt
done
echo "exit 0"
