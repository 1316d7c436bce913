#!/bin/bash
# round 5 visit 35: sf_igemm2 with 128 x 128 tiles (4 waves, 3 workgroups per CU) on the shallow token GEMMs
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v35; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
for V in "default:X=1" "igemm2 256x128 K>=384:SF_IGEMM2_MINK=384" "igemm2 128x128 K>=384:SF_IGEMM2_MINK=384,SF_IGEMM2_T128=1"; do
  L=${V%%:*}; E=${V#*:}
  echo "== $L" | tee -a $D/cold.txt
  env ${E//,/ } timeout 300 python tools/gemm_cold_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $D/cold.txt
done
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit base:X=1" "mvit t128 K>=384:SF_IGEMM2_MINK=384,SF_IGEMM2_T128=1" "mvit t128 K>=512:SF_IGEMM2_T128=1"
echo "exit 0"
