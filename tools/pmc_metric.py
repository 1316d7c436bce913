#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters / derived metrics (one CSV per pass) -> markdown table.

    python tools/pmc_metric.py out.md "title" <counter_collection.csv> [more.csv ...]

Every row of a counter_collection CSV is one (dispatch, counter) sample; the table lists, per kernel, the number of
dispatches and the launch-time-unweighted mean of every counter found (e.g. MfmaUtil, VALUBusy, SQ_VALU_MFMA_BUSY_CYCLES)."""
import collections
import csv
import sys


def main():
    out, title, paths = sys.argv[1], sys.argv[2], sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    counters = []
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                c = r.get("Counter_Name")
                if not c:
                    continue
                if c not in counters:
                    counters.append(c)
                a = agg[r["Kernel_Name"]][c]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    rows = []
    for k, cs in agg.items():
        n = max(v[0] for v in cs.values())
        rows.append((n, k, {c: (v[1] / v[0] if v[0] else float("nan")) for c, v in cs.items()}))
    rows.sort(key=lambda r: -r[0])
    with open(out, "w") as f:
        f.write(f"# {title}\n\nsource: rocprofv3 --pmc {' '.join(counters)} (per-dispatch samples averaged per kernel)\n\n")
        derived = "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "GRBM_GUI_ACTIVE" in counters
        if derived:
            f.write("MfmaUtil % = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 128): MFMA-busy cycles summed over the 1024 "
                    "SIMDs against the kernel's active cycles (GRBM_GUI_ACTIVE is the sum over the 8 XCDs: a 53 us kernel "
                    "reads ~1.0 M at 2.4 GHz), i.e. the gfx94x MfmaUtil formula written for 8 XCDs x 32 CUs x 4 SIMDs.\n\n")
        f.write("| kernel | dispatches | " + " | ".join(counters) + (" | MfmaUtil % |" if derived else " |") + "\n|---|---:|"
                + "---:|" * (len(counters) + int(derived)) + "\n")
        for n, k, m in rows[:40]:
            name = k if len(k) <= 90 else k[:87] + "..."
            line = f"| `{name}` | {n} | " + " | ".join(f"{m.get(c, float('nan')):.0f}" for c in counters)
            if derived:
                g = m.get("GRBM_GUI_ACTIVE", 0.0)
                line += f" | {100.0 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (g * 128.0):.1f}" if g else " | -"
            f.write(line + " |\n")
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main()
