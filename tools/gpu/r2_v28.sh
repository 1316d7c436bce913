#!/bin/bash
# round 2, visit 28: default bench line of the final code on another box (box-to-box spread) + smoke
mkdir -p gpurun_out/v28
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/v28/smoke.log 2>&1; echo "smoke rc=$?"
timeout 500 python bench.py > gpurun_out/v28/bench.log 2> gpurun_out/v28/bench.err; echo "bench rc=$?"; tail -1 gpurun_out/v28/bench.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['secondary']['value'], d['secondary']['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --no-kernel-profile --no-secondary > gpurun_out/v28/bench2.log 2>/dev/null; tail -1 gpurun_out/v28/bench2.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4
