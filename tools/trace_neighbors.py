"""Who launches the small device-to-device copies?  Reads a rocprofv3 --kernel-trace CSV (kernel_trace.csv), orders the
dispatches by start time and prints, for every dispatch whose kernel name contains `needle` (default copyBuffer), the kernels
right before and after it -- as a histogram of (previous, next) pairs.
    python tools/trace_neighbors.py <kernel_trace.csv> [needle] > profiles/r3/r3_copybuffer_neighbors.txt"""
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
            grid = "x".join(str(r.get(k, "?")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r \
                else str(r.get("Grid_Size", "?"))
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid))
    rows.sort()
    hist = collections.Counter()
    durs = collections.defaultdict(list)
    for i, (st, en, name, _g) in enumerate(rows):
        if needle in name:
            prev = short(rows[i - 1][2]) if i else "-"
            nxt = short(rows[i + 1][2]) if i + 1 < len(rows) else "-"
            hist[(prev, nxt)] += 1
            durs[(prev, nxt)].append((en - st) / 1e3)
    # the context of the longest run of matching dispatches: what precedes it, the sizes inside it, what follows
    best, cur, start = (0, 0), 0, 0
    for i, r in enumerate(rows):
        if needle in r[2]:
            if cur == 0:
                start = i
            cur += 1
            if cur > best[0]:
                best = (cur, start)
        else:
            cur = 0
    n, st0 = best
    if n:
        print(f"longest run: {n} dispatches starting at dispatch {st0}")
        for i in range(max(0, st0 - 6), st0):
            print("   before:", short(rows[i][2]), "grid", rows[i][3], "dur us %.1f" % ((rows[i][1] - rows[i][0]) / 1e3))
        sizes = collections.Counter(rows[i][3] for i in range(st0, st0 + n))
        print("   grid sizes inside the run:", sizes.most_common(12))
        for i in range(st0 + n, min(len(rows), st0 + n + 8)):
            print("   after: ", short(rows[i][2]), "grid", rows[i][3], "dur us %.1f" % ((rows[i][1] - rows[i][0]) / 1e3))
    # per-step count: dispatches matching the needle between consecutive optimizer launches (sf_flat_sgd / sf_flat_adamw)
    marks = [i for i, r in enumerate(rows) if "sf_flat_sgd" in r[2] or "sf_flat_adamw" in r[2]]
    for a, b in zip(marks, marks[1:]):
        n_in = sum(1 for i in range(a + 1, b) if needle in rows[i][2])
        t_in = sum((rows[i][1] - rows[i][0]) / 1e3 for i in range(a + 1, b) if needle in rows[i][2])
        print(f"step between optimizer launches at dispatch {a} and {b}: {b - a - 1} dispatches, {n_in} x {needle} ({t_in:.1f} us)")
    print(f"{sum(hist.values())} dispatches matching '{needle}' of {len(rows)}")
    for (prev, nxt), c in hist.most_common(40):
        d = durs[(prev, nxt)]
        print(f"{c:6d}  avg {sum(d) / len(d):6.2f} us   after [{prev}]   before [{nxt}]")


if __name__ == "__main__":
    main()
