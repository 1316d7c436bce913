#!/bin/bash
# round 2, visit 20: asm LDS-DMA in the weight-gradient kernels (no compiler-inserted vmcnt(0) before the transpose reads),
# thin weight-gradient kernel for <= 32 output channels.
mkdir -p gpurun_out/v20
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu --tb=short -k "igemm2 or wgrad" > gpurun_out/v20/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?"; tail -3 gpurun_out/v20/pytest_kernels.log | cut -c1-300
timeout 600 python tools/microbench.py --iters 5 --no-bn --md gpurun_out/v20/r2_v20_per_geometry.md > gpurun_out/v20/microbench.txt 2>&1; echo "microbench rc=$?"; grep "weighted totals" gpurun_out/v20/r2_v20_per_geometry.md
SF_WGRAD2T=0 timeout 600 python tools/microbench.py --iters 5 --no-bn --filter fast --md gpurun_out/v20/r2_v20_per_geometry_nothin.md > gpurun_out/v20/microbench_nothin.txt 2>&1; grep "weighted totals" gpurun_out/v20/r2_v20_per_geometry_nothin.md
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-secondary"
for P in "SLOWFAST_8x8_R50" "MVITv2_S_16x4"; do
  timeout 300 $B --preset $P > gpurun_out/v20/bench_$P.json 2> gpurun_out/v20/bench_$P.err; echo "$P: $(python -c "import json;d=json.loads(open('gpurun_out/v20/bench_$P.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
done
SF_WGRAD2T=0 timeout 300 $B > gpurun_out/v20/ab_nothin.json 2>/dev/null; echo "no thin kernel: $(python -c "import json;d=json.loads(open('gpurun_out/v20/ab_nothin.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])")"
timeout 900 python -m pytest tests/test_step.py tests/test_model_gpu.py -x -q -m gpu --tb=short > gpurun_out/v20/pytest_models.log 2>&1; echo "pytest models rc=$?"; tail -3 gpurun_out/v20/pytest_models.log | cut -c1-300
