#!/bin/bash
# round 2, visit 17: do Slow- and Fast-pathway kernels overlap across HIP streams / graph branches?
mkdir -p gpurun_out/v17
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 300 python tools/stream_overlap_probe.py > gpurun_out/v17/probe.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/v17/probe.txt | tail -14
