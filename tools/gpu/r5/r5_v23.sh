#!/bin/bash
# round 5 visit 23: where do the K = 384 token GEMMs spend their time?  igemm2 with parts switched off (diagnostic build), HBM-cold
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v23; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
export SFAMD_LIBRARY=$PWD/slowfast_amd/libsfamd_diag.so SF_IGEMM2_MINK=384
for V in "igemm2 bk32 full:SF_IGEMM2_BK=32" "igemm2 bk32 no-epilogue:SF_IGEMM2_BK=32,SF_IGEMM2_ABLATE=8" "igemm2 bk32 no-copies no-epilogue:SF_IGEMM2_BK=32,SF_IGEMM2_ABLATE=9" "igemm2 bk32 no-compute no-epilogue:SF_IGEMM2_BK=32,SF_IGEMM2_ABLATE=10" "igemm2 bk32 neither:SF_IGEMM2_BK=32,SF_IGEMM2_ABLATE=11" \
         "igemm2 bk64 full:SF_IGEMM2_BK=64" "igemm2 bk64 no-epilogue:SF_IGEMM2_BK=64,SF_IGEMM2_ABLATE=8" "igemm2 bk64 neither:SF_IGEMM2_BK=64,SF_IGEMM2_ABLATE=11"; do
  L=${V%%:*}; E=${V#*:}
  echo "== $L" | tee -a $D/cold.txt
  env ${E//,/ } timeout 300 python tools/gemm_cold_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-80 | tee -a $D/cold.txt
done
echo "exit 0"
