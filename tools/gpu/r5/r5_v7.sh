#!/bin/bash
# round 5 visit 7: batch-split probe (two half-batches on two streams) for the BatchNorm-free model
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v7; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 600 python tools/batch_split_probe.py 2>&1 | grep "ms per" | tee $D/probe.txt
timeout 600 python tools/batch_split_probe.py --splits 4 2>&1 | grep "ms per" | tee -a $D/probe.txt
echo "exit 0"
