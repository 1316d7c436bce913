#!/usr/bin/env python3
"""MFMA utilisation per layer and entry point from ONE rocprofv3 --pmc pass over `tools/microbench.py --markers --filter <layer>`.

    python tools/pmc_per_layer.py <counter_collection.csv> "<layer name>"   ->  one markdown table row on stdout

The microbenchmark launches a torch.arange kernel before its forward, data-gradient and weight-gradient phase and after the
last one; dispatches between two markers belong to one phase.  MfmaUtil % of a phase = sum(SQ_VALU_MFMA_BUSY_CYCLES) /
(sum(GRBM_GUI_ACTIVE) * 128) over the phase's convolution kernels (igemm / wgrad / stem; the fp32 split reduce, row-table and
fill kernels are left out), i.e. weighted by kernel time."""
import collections
import csv
import sys


def main():
    path, layer = sys.argv[1], sys.argv[2]
    disp = collections.OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    phases, cur = [], None
    for k in sorted(disp):
        d = disp[k]
        if "arange" in d["name"]:
            cur = []
            phases.append(cur)
        elif cur is not None:
            cur.append(d)
    cells = []
    for ph in phases[:3]:
        conv = [d for d in ph if any(t in d["name"] for t in ("igemm", "wgrad2_kernel", "wgrad2t_kernel", "wgrad_kernel", "stem_"))]
        g = sum(d.get("GRBM_GUI_ACTIVE", 0.0) for d in conv)
        m = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in conv)
        names = sorted({d["name"].split("(")[0].replace("void ", "") for d in conv})
        cells.append((f"{100.0 * m / (g * 128.0):.1f}" if g else "-", ", ".join(n[:44] for n in names)))
    while len(cells) < 3:
        cells.append(("-", ""))
    print(f"| {layer} | " + " | ".join(c[0] for c in cells) + " | " + " / ".join(c[1] for c in cells) + " |")


if __name__ == "__main__":
    main()
