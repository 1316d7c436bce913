#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats` run into a small text table under profiles/.

rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default; its `top_kernels` view is the per-kernel
statistics table.  Usage: python tools/rocprof_summary.py gpurun_out/prof/r1_results.db profiles/r1_xxx.md "title"
Works on CSV output too (`*_kernel_stats.csv`)."""
import csv
import sqlite3
import sys


def rows_from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    return [(n, int(c), float(t), float(a), float(p)) for n, c, t, a, p in
            cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels")]


def rows_from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                        float(r["Percentage"])))
    return out


def short(name):
    name = name.replace("void ", "")
    if len(name) > 90:
        name = name[:87] + "..."
    return name


def main():
    src, dst = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else src
    rows = rows_from_db(src) if src.endswith(".db") else rows_from_csv(src)
    scale = 1.0
    if src.endswith(".db") and rows and rows[0][3] > 0:
        # rocpd durations are in ns; print microseconds
        scale = 1e-3 if max(r[3] for r in rows) > 1e4 else 1.0
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    with open(dst, "w") as f:
        f.write(f"# {title}\n\nsource: rocprofv3 --kernel-trace --stats ({src.split('/')[-1]}); durations in us\n\n")
        f.write(f"total kernel time: {tot * scale:.1f} us over {sum(r[1] for r in rows)} dispatches\n\n")
        f.write("| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for n, c, t, a, p in rows[:40]:
            f.write(f"| `{short(n)}` | {c} | {t * scale:.1f} | {a * scale:.2f} | {100.0 * t / tot:.2f} |\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()
