#!/usr/bin/env python3
"""Collect the one-line JSON results of tools/gpu/all_presets.sh into one file: python tools/collect_presets.py <dir> <out.json>"""
import glob
import json
import os
import sys

out = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        out[os.path.basename(f)[:-5]] = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:      # a preset that failed keeps its error text
        out[os.path.basename(f)[:-5]] = {"error": str(e)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
