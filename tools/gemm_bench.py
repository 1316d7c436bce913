"""The nn.Linear shapes of MViTv2-S (16x224^2, batch 32) through tokens.gemm / gemm_gelu / linear_wgrad: us per call, algorithmic
GB/s and TFLOP/s per shape (HIP events around `iters` back-to-back calls; operands re-created per shape).
    python tools/gemm_bench.py [--md gpurun_out/x/gemm_bench.md] [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slowfast_amd import lib, tokens  # noqa: E402
from tools.microbench import timeit  # noqa: E402

# (stage, tokens per clip, dim, dim_out of the stage's last block, blocks)
STAGES = [("s1", 25089, 96, 192, 1), ("s2", 6273, 192, 384, 2), ("s3", 1569, 384, 768, 11), ("s4", 393, 768, 768, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--md", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = lib.act_dtype()
    lines = ["| layer | x | M | K | N | fwd us | GB/s | TF | dgrad us | GB/s | wgrad us | GB/s |", "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    tot = [0.0, 0.0, 0.0]
    for st, ntok, dim, dim_out, blocks in STAGES:
        M = a.batch * ntok
        for name, K, N, cnt in (("qkv", dim, 3 * dim, blocks), ("proj", dim, dim, blocks), ("fc1", dim, 4 * dim, blocks),
                                ("fc2", 4 * dim, dim, blocks - 1), ("fc2 last", 4 * dim, dim_out, 1)):
            if cnt == 0:
                continue
            x = torch.randn((M, K), device=dev).to(dt)
            w = (torch.randn((N, K), device=dev) * 0.05).to(dt)
            wt = w.t().contiguous()
            dy = torch.randn((M, N), device=dev).to(dt)
            y = torch.empty((M, N), dtype=dt, device=dev)
            dx = torch.empty((M, K), dtype=dt, device=dev)
            dw = torch.empty((N, K), device=dev)
            bias = torch.zeros(N, device=dev)
            t_f = timeit(lambda: tokens.gemm(x, w, bias=bias, out=y), a.iters) * 1e3
            t_d = timeit(lambda: tokens.gemm(dy, wt, out=dx), a.iters) * 1e3
            t_w = timeit(lambda: tokens.linear_wgrad(x, dy, dw), a.iters) * 1e3
            by = 2.0 * (M * K + M * N)
            fl = 2.0 * M * K * N
            lines.append(f"| {st} {name} | {cnt} | {M} | {K} | {N} | {t_f:.0f} | {by / t_f / 1e3:.0f} | {fl / t_f / 1e6:.0f} | {t_d:.0f} | {by / t_d / 1e3:.0f} | "
                         f"{t_w:.0f} | {by / t_w / 1e3:.0f} |")
            print(lines[-1], flush=True)
            tot[0] += cnt * t_f; tot[1] += cnt * t_d; tot[2] += cnt * t_w
    lines.append(f"| **weighted total (ms / step)** | | | | | {tot[0] / 1e3:.2f} | | | {tot[1] / 1e3:.2f} | | {tot[2] / 1e3:.2f} | |")
    print(lines[-1])
    if a.md:
        os.makedirs(os.path.dirname(a.md) or ".", exist_ok=True)
        with open(a.md, "w") as f:
            f.write("# nn.Linear shapes of MViTv2-S 16x224^2, batch %d: tokens.gemm (fwd, + bias) / gemm (data gradient) / linear_wgrad, us per call; "
                    "GB/s = (M*K + M*N) * 2 B / time\n\n" % a.batch)
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
