#!/bin/bash
# round 5 visit 51: the default bench line at HEAD
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/v51; mkdir -p $D; export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 200 python bench.py --no-cpu-baseline > $D/bench.log 2> $D/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$D/bench.log').read().strip().split('\n')[-1]); print('value', d['value'], d['ms_per_step'], 'secondary', d['secondary']['value'], d['secondary']['ms_per_step'])"
echo "exit 0"
