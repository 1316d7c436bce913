#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
SECONDS=0
timeout 900 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "bench default rc=$? wall=${SECONDS}s"; tail -1 gpurun_out/bench_default.log | cut -c1-3500
