#!/usr/bin/env python3
"""HBM-cold timing of the MViTv2-S stage-3 token GEMMs: every call works on its OWN operand / output buffers, cycling through
enough sets (> 1 GB) that nothing comes out of the 256 MB Infinity Cache -- the situation inside the training step, which a
back-to-back microbenchmark on one buffer set hides (fc1 forward 88 us warm, 146 us in the step).
    python tools/gemm_cold_bench.py [--sets 6] [--iters 3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slowfast_amd import lib, tokens  # noqa: E402


def cold(fn, sets, iters):
    for s in sets:
        fn(s)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        for s in sets:
            fn(s)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (iters * len(sets))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=6)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    dev, dt = torch.device("cuda:0"), lib.act_dtype()
    M, C = 32 * 1569, 384
    for name, K, N in (("qkv", C, 3 * C), ("proj", C, C), ("fc1", C, 4 * C), ("fc2", 4 * C, C)):
        sets = []
        for _ in range(a.sets):
            sets.append(dict(x=torch.randn((M, K), device=dev).to(dt), w=(torch.randn((N, K), device=dev) * 0.05).to(dt),
                             y=torch.empty((M, N), dtype=dt, device=dev), bias=torch.zeros(N, device=dev),
                             h=torch.randn((M, N), device=dev).to(dt) if name == "fc1" else None))
        by = 2.0 * (M * K + M * N)
        t = cold(lambda s: tokens.gemm(s["x"], s["w"], bias=s["bias"], out=s["y"]), sets, a.iters)
        line = f"{name:5s} M={M} K={K} N={N}: plain {t:6.1f} us {by / t / 1e3:5.0f} GB/s {2.0 * M * N * K / t / 1e6:5.0f} TF"
        if name == "fc1":
            t2 = cold(lambda s: tokens.gemm_gelu(s["x"], s["w"], bias=s["bias"]), sets, a.iters)
            line += f" | +gelu (two outputs, allocator) {t2:6.1f} us {(by + 2.0 * M * N) / t2 / 1e3:5.0f} GB/s"
        if name == "fc2":       # its data gradient: [M, C] x [C -> 4C] with the gelu' epilogue reading h
            wt = [torch.randn((K, N), device=dev).to(dt) * 0.05 for _ in sets]
            hs = [torch.randn((M, K), device=dev).to(dt) for _ in sets]
            dys = [torch.randn((M, N), device=dev).to(dt) for _ in sets]
            idx = list(range(len(sets)))
            t3 = cold(lambda i: tokens.gemm_gelu_grad(dys[i], wt[i], hs[i]), idx, a.iters)
            line += f" | dgrad x gelu' {t3:6.1f} us {2.0 * (M * N + 2 * M * K) / t3 / 1e3:5.0f} GB/s"
        print(line, flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
