#!/bin/bash
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v13; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 600 python tools/batch_split_probe.py > $D/probe_full.txt 2>&1; grep -E "ms per" $D/probe_full.txt; tail -12 $D/probe_full.txt | cut -c1-250
echo "exit 0"
