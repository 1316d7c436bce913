"""Per-kernel timing of libsfamd on the SlowFast-8x8-R50 layer geometries.

Times conv fwd / dgrad / wgrad and the BN/elementwise kernels with HIP events on torch's current
stream, prints achieved TFLOP/s and algorithmic GB/s per layer.  Usage:
    python tools/microbench.py [--batch 32] [--iters 5] [--json gpurun_out/microbench.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slowfast_amd import ops  # noqa: E402

# (name, Ci, T, H, W, Co, kernel, stride, pad, count)
LAYERS = [
    # RGB stems as W-pair-folded convolutions (engine.StemConvUnit): input (N,8,T,224,112), kernel (kT,7,4)
    ("slow.stem 3->64 1x7x7/2", 8, 8, 224, 112, 64, (1, 7, 4), (1, 2, 1), (0, 3, 2), 1),
    ("fast.stem 3->8 5x7x7/2", 8, 32, 224, 112, 8, (5, 7, 4), (1, 2, 1), (2, 3, 2), 1),
    ("s2.slow a 80->64 1x1", 80, 8, 56, 56, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), 1),
    ("s2.slow b 64->64 1x3x3", 64, 8, 56, 56, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), 3),
    ("s2.slow c 64->256 1x1", 64, 8, 56, 56, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), 3),
    ("s2.slow a 256->64 1x1", 256, 8, 56, 56, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), 2),
    ("s2.fast a 32->8 3x1x1", 32, 32, 56, 56, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), 2),
    ("s2.fast b 8->8 1x3x3", 8, 32, 56, 56, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), 3),
    ("s2.fast c 8->32 1x1", 8, 32, 56, 56, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), 4),
    ("fuse2 32->64 7x1x1/4", 32, 32, 56, 56, 64, (7, 1, 1), (4, 1, 1), (3, 0, 0), 1),
    ("s3.slow a 320->128 1x1", 320, 8, 56, 56, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), 1),
    ("s3.slow b 128->128 3x3/2", 128, 8, 56, 56, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), 1),
    ("s3.slow sc 320->512 1x1/2", 320, 8, 56, 56, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), 1),
    ("s3.slow b 128->128 3x3", 128, 8, 28, 28, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), 3),
    ("s3.slow c 128->512 1x1", 128, 8, 28, 28, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), 4),
    ("s3.slow a 512->128 1x1", 512, 8, 28, 28, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), 3),
    ("s3.fast a 64->16 3x1x1", 64, 32, 28, 28, 16, (3, 1, 1), (1, 1, 1), (1, 0, 0), 3),
    ("s3.fast c 16->64 1x1", 16, 32, 28, 28, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), 4),
    ("s4.slow a 640->256 3x1x1", 640, 8, 28, 28, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), 1),
    ("s4.slow a 1024->256 3x1x1", 1024, 8, 14, 14, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), 5),
    ("s4.slow b 256->256 3x3", 256, 8, 14, 14, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), 5),
    ("s4.slow c 256->1024 1x1", 256, 8, 14, 14, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), 6),
    ("s4.fast a 128->32 3x1x1", 128, 32, 14, 14, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), 5),
    ("s5.slow a 1280->512 3x1x1", 1280, 8, 14, 14, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), 1),
    ("s5.slow a 2048->512 3x1x1", 2048, 8, 7, 7, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), 2),
    ("s5.slow b 512->512 3x3", 512, 8, 7, 7, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), 2),
    ("s5.slow c 512->2048 1x1", 512, 8, 7, 7, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0), 3),
]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


HBM_PEAK_GBS, MFMA_PEAK_TF = 8000.0, 2500.0


def write_markdown(path, batch, rows, tot):
    """Per-geometry table: microseconds, algorithmic GB/s (operand + result bytes crossing HBM once), TFLOP/s and the
    roofline fraction max(GB/s / 8 TB/s, TFLOP/s / 2.5 PFLOP/s) of each entry point; worst shapes by time lost."""
    def frac(gbs, tf):
        return max(gbs / HBM_PEAK_GBS, tf / MFMA_PEAK_TF)
    lines = [f"# libsfamd entry points on the SlowFast-8x8-R50 layer geometries, batch {batch} (tools/microbench.py, HIP events)", "",
             "us per call; GB/s = algorithmic bytes (input + output activations once, fp16) / time; TF = useful TFLOP/s; "
             "frac = max(GB/s / 8000, TF / 2500) = distance to whichever roofline bounds the layer", "",
             "| layer | x | fwd us | GB/s | TF | frac | dgrad us | GB/s | TF | frac | wgrad us | GB/s | TF | frac | bn_act us | GB/s | bn_bwd us | GB/s |",
             "|---|---:|" + "---:|" * 16]
    lost = []
    for r in rows:
        io = r["io_bytes"]
        cells = [r["name"], str(r["count"])]
        for op in ("fwd", "dgrad", "wgrad"):
            ms = r[f"{op}_ms"]
            if not ms:
                cells += ["-", "-", "-", "-"]
                continue
            gbs, tf = io / ms / 1e6, r[f"{op}_tflops"]
            f = frac(gbs, tf)
            cells += [f"{ms * 1e3:.0f}", f"{gbs:.0f}", f"{tf:.0f}", f"{f:.2f}"]
            lost.append((r["count"] * ms * (1 - f), f"{r['name']} {op}", r["count"], ms, gbs, tf, f))
        cells += [f"{r['bn_act_ms'] * 1e3:.0f}", f"{r['act_gbs']:.0f}", f"{r['bn_bwd_ms'] * 1e3:.0f}", f"{r['bnbwd_gbs']:.0f}"]
        lines.append("| " + " | ".join(cells) + " |")
    lines += ["", "weighted totals per step (ms): " + ", ".join(f"{k} {v:.2f}" for k, v in tot.items()), "",
              "## Worst shapes (count x time x (1 - frac) = ms per step above the roofline)", "",
              "| layer / op | x | us | GB/s | TF | frac | ms lost |", "|---|---:|---:|---:|---:|---:|---:|"]
    for l, name, cnt, ms, gbs, tf, f in sorted(lost, reverse=True)[:15]:
        lines.append(f"| {name} | {cnt} | {ms * 1e3:.0f} | {gbs:.0f} | {tf:.0f} | {f:.2f} | {l:.2f} |")
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--json", default="")
    ap.add_argument("--filter", default="", help="substring of the layer name; several alternatives separated by |")
    ap.add_argument("--no-bn", action="store_true", help="skip the BatchNorm / activation kernels (GEMM variant sweeps)")
    ap.add_argument("--markers", action="store_true", help="launch a torch.arange kernel before the fwd / dgrad / wgrad phase of "
                    "every layer and after the last one: phase separators for tools/pmc_per_layer.py under rocprofv3 --pmc")
    ap.add_argument("--md", default="", help="write the per-geometry table (us, GB/s, TFLOP/s, roofline fraction) as markdown")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "bn": 0.0}
    for name, Ci, T, H, W, Co, k, s, p, cnt in LAYERS:
        if a.filter and not any(f in name for f in a.filter.split("|")):
            continue
        Cw = Ci
        stem = "stem" in name
        geom = ops.ConvGeom((a.batch, Ci, T, H, W), Co, k, s, p, Cw=Cw, out_dims=(T, 112, 112) if stem else None)
        x = ops.cl_empty(geom.in_shape, dev)
        x.normal_()
        w = torch.randn((Co, Cw) + k, device=dev) * 0.05
        wf, wd = ops.prep_weights(w, geom)
        y, part = ops.conv_fwd(x, wf, geom)
        dy = ops.cl_empty(geom.out_shape, dev)
        dy.normal_()
        dw = torch.empty_like(w)
        sc = torch.ones(Ci, device=dev)
        sh = torch.zeros(Ci, device=dev)
        macs = geom.out_rows * Co * (3 * k[0] * 49 if stem else Cw * geom.taps)   # useful MACs
        mark = (lambda: torch.arange(16, device=dev)) if a.markers else (lambda: None)
        mark()
        t_f = timeit(lambda: ops.conv_fwd(x, wf, geom, out=y), a.iters)
        mark()
        t_fa = timeit(lambda: ops.conv_fwd(x, wf, geom, in_affine=(sc, sh, True), out=y), a.iters) if Ci <= 512 and not a.no_bn else float("nan")
        dx = ops.cl_empty(geom.in_shape, dev)
        t_d = timeit(lambda: ops.conv_dgrad(dy, wd, geom, out=dx), a.iters) if "stem" not in name else 0.0
        mark()
        t_w = timeit(lambda: ops.conv_wgrad(x, dy, geom, dw), a.iters)
        mark()
        gamma = torch.ones(Co, device=dev); beta = torch.zeros(Co, device=dev)
        rm = torch.zeros(Co, device=dev); rv = torch.ones(Co, device=dev)
        scale, shift, mean, rstd = ops.bn_finalize(part, geom.out_rows, gamma, beta, rm, rv, 0.1, 1e-5)
        z = ops.cl_empty(geom.out_shape, dev)
        dgm = torch.empty(Co, device=dev); dbt = torch.empty(Co, device=dev)
        if a.no_bn:
            t_act = t_bb = float("nan")
        else:
            t_act = timeit(lambda: ops.bn_act(y, scale, shift, relu=True, out=z), a.iters)
            t_bb = timeit(lambda: ops.bn_bwd(dy, y, gamma, mean, rstd, dgm, dbt, relu_affine=(scale, shift), out=z), a.iters)
        bytes_io = 2.0 * (x.numel() * Cw / Ci + y.numel())
        row = dict(name=name, count=cnt, gmac=macs / 1e9, io_bytes=bytes_io, fwd_ms=t_f, fwd_affine_ms=t_fa, dgrad_ms=t_d, wgrad_ms=t_w,
                   bn_act_ms=t_act, bn_bwd_ms=t_bb, fwd_tflops=2 * macs / t_f / 1e9,
                   dgrad_tflops=(2 * macs / t_d / 1e9 if t_d else 0), wgrad_tflops=2 * macs / t_w / 1e9,
                   fwd_gbs=bytes_io / t_f / 1e6, act_gbs=2.0 * 2 * y.numel() / t_act / 1e6,
                   bnbwd_gbs=2.0 * 5 * y.numel() / t_bb / 1e6)
        rows.append(row)
        tot["fwd"] += cnt * t_f; tot["dgrad"] += cnt * t_d; tot["wgrad"] += cnt * t_w; tot["bn"] += 0.0 if a.no_bn else cnt * (t_act + t_bb)
        print(f"{name:30s} x{cnt} {macs/1e9:7.1f} GMAC | fwd {t_f:7.3f} ms {row['fwd_tflops']:7.1f} TF {row['fwd_gbs']:7.0f} GB/s "
              f"(+bn {t_fa:7.3f}) | dgrad {t_d:7.3f} ms {row['dgrad_tflops']:7.1f} TF | wgrad {t_w:7.3f} ms {row['wgrad_tflops']:7.1f} TF "
              f"| act {t_act:6.3f} ms {row['act_gbs']:6.0f} GB/s | bnbwd {t_bb:6.3f} ms {row['bnbwd_gbs']:6.0f} GB/s", flush=True)
    print("weighted totals (ms):", {k: round(v, 2) for k, v in tot.items()}, "sum", round(sum(tot.values()), 2))
    if a.md:
        write_markdown(a.md, a.batch, rows, tot)
    if a.json:
        os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(dict(batch=a.batch, rows=rows, totals=tot), f, indent=1)


if __name__ == "__main__":
    main()
