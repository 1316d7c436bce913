#!/bin/bash
# round 5 visit 25: sf_igemm2p with EIGHT drain waves
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v25; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest -q -m gpu -x --tb=short tests/test_igemm2p_gpu.py > $D/pytest.log 2>&1; echo "pytest igemm2p rc=$?"; tail -4 $D/pytest.log | cut -c1-300
for V in "default:X=1" "igemm2p:SF_IGEMM2P=1"; do
  L=${V%%:*}; E=${V#*:}
  echo "== $L" | tee -a $D/cold.txt
  env ${E//,/ } timeout 300 python tools/gemm_cold_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $D/cold.txt
done
ROUNDS=2 bash tools/gpu/ab.sh $D --preset MVITv2_S_16x4 -- "mvit base:X=1" "mvit igemm2p>=512:SF_IGEMM2P=512" "mvit igemm2p>=512 mink384:SF_IGEMM2P=512,SF_IGEMM2P_MINK=384"
echo "exit 0"
