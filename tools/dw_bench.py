"""Microbenchmark of the depthwise 3x3x3 convolutions at the MViTv2-S (batch 32) and X3D-M (batch 64) shapes: forward, data
gradient, weight gradient, HIP-event timed with COLD operands (the calls rotate through enough buffer sets to exceed the
256 MiB Infinity Cache), the ring sweep (sf_dwsweep.h) against the kernels it replaces (SF_DW_SWEEP=0) in one process.
`python tools/dw_bench.py [--iters N] [--only mvit|x3d] [--env K=V,K=V]`"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from slowfast_amd import lib, tokens


def timed(fns, iters):
    for f in fns:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fns[i % len(fns)]()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def dw_case(name, B, heads, Cw, thw, stride, cls, iters, dev, stats=False, ld_mult=1):
    f16 = lib.act_dtype()
    C = heads * Cw
    geom = tokens.DwGeom(B, C, Cw, thw, (3, 3, 3), stride, (1, 1, 1), cls)
    g = torch.Generator().manual_seed(0)
    by_in, by_out = 2.0 * C * geom.rows_in, 2.0 * C * geom.rows_out
    nset = max(2, int(400e6 // (by_in * ld_mult + by_out)) + 1)
    xs, dys = [], []
    for _ in range(nset):
        big = torch.randn((geom.rows_in, C * ld_mult), generator=g).to(f16).to(dev)
        xs.append(big[:, :C] if ld_mult > 1 else big)          # a channel slice of a wider tensor (the q / k / v slices of qkv)
        dys.append(torch.randn((geom.rows_out, C), generator=g).to(f16).to(dev))
    w = torch.randn((Cw, 1, 3, 3, 3), generator=g).to(dev)
    dw = torch.empty_like(w)
    res = {}
    for label, sweep in (("old", "0"), ("new", "1")):
        os.environ["SF_DW_SWEEP"] = sweep
        t_f = timed([(lambda x=x: tokens.dwconv_fwd(x, w, geom, stats=stats)) for x in xs], iters)
        t_d = timed([(lambda dy=dy: tokens.dwconv_dgrad(dy, w, geom)) for dy in dys], iters)
        t_w = timed([(lambda x=x, dy=dy: tokens.dwconv_wgrad(x, dy, geom, dw)) for x, dy in zip(xs, dys)], iters)
        res[label] = (t_f, t_d, t_w)
    by = by_in + by_out
    o, n_ = res["old"], res["new"]
    print(f"{name:34s} MB {by * 1e-6:7.1f} | fwd {o[0]:7.1f} -> {n_[0]:7.1f} us ({by / n_[0] * 1e-3:5.0f} GB/s) | dgrad {o[1]:7.1f} -> "
          f"{n_[1]:7.1f} ({by / n_[1] * 1e-3:5.0f}) | wgrad {o[2]:7.1f} -> {n_[2]:7.1f} ({by / n_[2] * 1e-3:5.0f})", flush=True)
    return by, o, n_


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--only", default="")
    ap.add_argument("--env", default="")
    a = ap.parse_args()
    for kv in a.env.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1)
            os.environ[k] = v
    dev = torch.device("cuda:0")
    tot_o, tot_n = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]

    def run(count, *args, **kw):
        _, o, n_ = dw_case(*args, iters=a.iters, dev=dev, **kw)
        for i in range(3):
            tot_o[i] += count * o[i]
            tot_n[i] += count * n_[i]

    if a.only == "s3":         # the two stage-3 shapes alone (what tools/gpu/pmc_dw.sh points rocprofv3 at)
        run(10, "mvit s3 q 14x14 s1 C384", 32, 4, 96, (8, 14, 14), (1, 1, 1), 1, ld_mult=3)
        run(20, "mvit s3 k/v 14->7 s2 C384", 32, 4, 96, (8, 14, 14), (1, 2, 2), 1, ld_mult=3)
    if a.only == "probe":      # fixed cost vs per-plane cost of a sweep: the stage-3 shapes at T = 4 / 8 / 16 / 32
        for T in (4, 8, 16, 32):
            run(0, f"probe s3 q s1 T={T}", 32, 4, 96, (T, 14, 14), (1, 1, 1), 1, ld_mult=3)
            run(0, f"probe s3 kv s2 T={T}", 32, 4, 96, (T, 14, 14), (1, 2, 2), 1, ld_mult=3)
    if a.only == "probe2":     # residency: 504 / 768 / 1008 / 1536 workgroups of the stage-3 stride-1 sweep
        for B in (10, 21, 32, 42, 64):
            run(0, f"probe s3 q s1 B={B}", B, 4, 96, (8, 14, 14), (1, 1, 1), 1, ld_mult=3)
    if a.only in ("", "mvit"):
        # (launches per step, name, B, heads, Cw, thw, stride, cls); the k / v pools read slices of the 3C-wide qkv tensor
        run(2, "mvit b0 k/v 56->7 s8 C96", 32, 1, 96, (8, 56, 56), (1, 8, 8), 1, ld_mult=3)
        run(2, "mvit b1 k/v 56->14 s4 C192", 32, 2, 96, (8, 56, 56), (1, 4, 4), 1, ld_mult=3)
        run(2, "mvit b2 k/v 28->7 s4 C192", 32, 2, 96, (8, 28, 28), (1, 4, 4), 1, ld_mult=3)
        run(1, "mvit b0 q 56x56 s1 C96", 32, 1, 96, (8, 56, 56), (1, 1, 1), 1, ld_mult=3)
        run(1, "mvit b1 q 56->28 s2 C192", 32, 2, 96, (8, 56, 56), (1, 2, 2), 1, ld_mult=3)
        run(1, "mvit b2 q 28x28 s1 C192", 32, 2, 96, (8, 28, 28), (1, 1, 1), 1, ld_mult=3)
        run(3, "mvit b3 q/k/v 28->14 s2 C384", 32, 4, 96, (8, 28, 28), (1, 2, 2), 1, ld_mult=3)
        run(10, "mvit s3 q 14x14 s1 C384", 32, 4, 96, (8, 14, 14), (1, 1, 1), 1, ld_mult=3)
        run(20, "mvit s3 k/v 14->7 s2 C384", 32, 4, 96, (8, 14, 14), (1, 2, 2), 1, ld_mult=3)
        run(1, "mvit b14 q 14->7 s2 C768", 32, 8, 96, (8, 14, 14), (1, 2, 2), 1, ld_mult=3)
        run(2, "mvit b14 k/v 14x14 s1 C768", 32, 8, 96, (8, 14, 14), (1, 1, 1), 1, ld_mult=3)
        run(3, "mvit b15 q/k/v 7x7 s1 C768", 32, 8, 96, (8, 7, 7), (1, 1, 1), 1, ld_mult=3)
        print(f"MViTv2-S per step (launch-weighted): fwd {tot_o[0] * 1e-3:.2f} -> {tot_n[0] * 1e-3:.2f} ms, dgrad {tot_o[1] * 1e-3:.2f} -> "
              f"{tot_n[1] * 1e-3:.2f}, wgrad {tot_o[2] * 1e-3:.2f} -> {tot_n[2] * 1e-3:.2f}", flush=True)
        tot_o[:] = [0.0, 0.0, 0.0]
        tot_n[:] = [0.0, 0.0, 0.0]
    if a.only in ("", "x3d"):
        run(1, "x3d s2 112->56 s2 C56", 64, 1, 56, (16, 112, 112), (1, 2, 2), 0, stats=True)
        run(2, "x3d s2 56x56 s1 C56", 64, 1, 56, (16, 56, 56), (1, 1, 1), 0, stats=True)
        run(1, "x3d s3 56->28 s2 C112", 64, 1, 112, (16, 56, 56), (1, 2, 2), 0, stats=True)
        run(4, "x3d s3 28x28 s1 C112", 64, 1, 112, (16, 28, 28), (1, 1, 1), 0, stats=True)
        run(1, "x3d s4 28->14 s2 C216", 64, 1, 216, (16, 28, 28), (1, 2, 2), 0, stats=True)
        run(10, "x3d s4 14x14 s1 C216", 64, 1, 216, (16, 14, 14), (1, 1, 1), 0, stats=True)
        run(1, "x3d s5 14->7 s2 C432", 64, 1, 432, (16, 14, 14), (1, 2, 2), 0, stats=True)
        run(6, "x3d s5 7x7 s1 C432", 64, 1, 432, (16, 7, 7), (1, 1, 1), 0, stats=True)
        print(f"X3D-M per step (launch-weighted): fwd {tot_o[0] * 1e-3:.2f} -> {tot_n[0] * 1e-3:.2f} ms, dgrad {tot_o[1] * 1e-3:.2f} -> "
              f"{tot_n[1] * 1e-3:.2f}, wgrad {tot_o[2] * 1e-3:.2f} -> {tot_n[2] * 1e-3:.2f}", flush=True)


if __name__ == "__main__":
    main()
