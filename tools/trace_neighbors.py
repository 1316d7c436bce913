"""Who launches the small device-to-device copies?  Reads a rocprofv3 --kernel-trace CSV (kernel_trace.csv), orders the
dispatches by start time and prints, for every dispatch whose kernel name contains `needle` (default copyBuffer), the kernels
right before and after it -- as a histogram of (previous, next) pairs.
    python tools/trace_neighbors.py <kernel_trace.csv> [needle] > profiles/r3_copybuffer_neighbors.txt"""
import collections
import csv
import sys


def short(n):
    n = n.replace("void ", "")
    i = n.find("(")
    n = n if i < 0 else n[:i]
    return n[:70]


def main():
    path, needle = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "copyBuffer")
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    hist = collections.Counter()
    durs = collections.defaultdict(list)
    for i, (st, en, name) in enumerate(rows):
        if needle in name:
            prev = short(rows[i - 1][2]) if i else "-"
            nxt = short(rows[i + 1][2]) if i + 1 < len(rows) else "-"
            hist[(prev, nxt)] += 1
            durs[(prev, nxt)].append((en - st) / 1e3)
    print(f"{sum(hist.values())} dispatches matching '{needle}' of {len(rows)}")
    for (prev, nxt), c in hist.most_common(40):
        d = durs[(prev, nxt)]
        print(f"{c:6d}  avg {sum(d) / len(d):6.2f} us   after [{prev}]   before [{nxt}]")


if __name__ == "__main__":
    main()
