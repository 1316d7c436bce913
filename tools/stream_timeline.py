#!/usr/bin/env python3
"""Per-queue timeline of a rocprofv3 --kernel-trace CSV: how much of the traced window has 0 / 1 / 2+ HIP queues busy, and which
kernels run while the OTHER queues are idle (the exposed part of each stream).  Written for engine.run_pathways (Slow / Fast
pathway of SlowFast on two streams): shows which pathway the joins wait for.
    python tools/stream_timeline.py <kernel_trace.csv> [out.md] [--last-frac 0.5]"""
import csv
import sys
from collections import defaultdict


def main():
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    frac = 0.5
    if "--last-frac" in sys.argv:
        frac = float(sys.argv[sys.argv.index("--last-frac") + 1])
    rows = []
    with open(src) as f:
        for r in csv.DictReader(f):
            q = r.get("Queue_Id") or r.get("Stream_Id") or "0"
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), q, r["Kernel_Name"]))
    rows.sort()
    # the timed steps are the LAST dense cluster of dispatches: cut at the last idle gap longer than 2 ms (model build, graph
    # capture and the warm-up iterations sit before it); --last-frac keeps at most that share of the trace
    t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
    cut = t_hi - (t_hi - t_lo) * frac
    end = rows[0][1]
    for s_, e_, _, _ in rows:
        if s_ - end > 2_000_000 and s_ > cut:
            cut = s_
        end = max(end, e_)
    rows = [r for r in rows if r[0] >= cut]
    t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, q, n in rows:
        ev.append((s, 1, q, n))
        ev.append((e, -1, q, n))
    ev.sort(key=lambda x: (x[0], x[1]))
    active = defaultdict(int)
    names = {}
    busy_by_count = defaultdict(int)
    alone_by_queue = defaultdict(int)
    alone_kernel = defaultdict(int)
    last = t_lo
    for t, d, q, n in ev:
        nb = sum(1 for v in active.values() if v > 0)
        dt = t - last
        busy_by_count[nb] += dt
        if nb == 1:
            (qq,) = [k for k, v in active.items() if v > 0]
            alone_by_queue[qq] += dt
            alone_kernel[(qq, names.get(qq, "?"))] += dt
        last = t
        active[q] += d
        if d > 0:
            names[q] = n
    wall = t_hi - t_lo
    per_queue = defaultdict(int)
    cnt = defaultdict(int)
    for s, e, q, n in rows:
        per_queue[q] += e - s
        cnt[q] += 1
    out = [f"# queue timeline of {src} (last dense cluster of the trace: {wall / 1e6:.2f} ms, {len(rows)} dispatches)", ""]
    out.append("| queues busy | ms | share |")
    out.append("|---:|---:|---:|")
    for k in sorted(busy_by_count):
        out.append(f"| {k} | {busy_by_count[k] / 1e6:.2f} | {busy_by_count[k] / wall:.1%} |")
    out += ["", "| queue | dispatches | sum of kernel durations ms | ms with every other queue idle |", "|---|---:|---:|---:|"]
    for q in sorted(per_queue, key=lambda k: -per_queue[k]):
        out.append(f"| {q} | {cnt[q]} | {per_queue[q] / 1e6:.2f} | {alone_by_queue[q] / 1e6:.2f} |")
    out += ["", "Kernels running while every other queue is idle (top 25 by time):", "", "| queue | kernel | ms alone |", "|---|---|---:|"]
    for (q, n), v in sorted(alone_kernel.items(), key=lambda kv: -kv[1])[:25]:
        out.append(f"| {q} | `{n[:100]}` | {v / 1e6:.2f} |")
    text = "\n".join(out) + "\n"
    if dst:
        open(dst, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
