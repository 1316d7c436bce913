"""Microbenchmark of the MViT token-space kernels at MViTv2-S shapes (batch 32): fused attention forward / backward and the
depthwise pooling convolutions, HIP-event timed; also the thing `rocprofv3 --pmc` is pointed at (tools/gpu/r4_v3.sh).
`python tools/token_bench.py [--iters N] [--only attn|dw|ln|colsum|stage3]`"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from slowfast_amd import lib, tokens


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def attn_case(B, heads, D, q_thw, k_thw, iters, dev):
    f16 = lib.act_dtype()
    C = heads * D
    d = tokens.attn_desc(B, heads, D, 1, q_thw, k_thw, 2 * max(q_thw[1], k_thw[1]) - 1, 2 * max(q_thw[2], k_thw[2]) - 1,
                         2 * max(q_thw[0], k_thw[0]) - 1)
    g = torch.Generator().manual_seed(0)
    q = torch.randn((B, d.Nq, C), generator=g).to(f16).to(dev)
    k = torch.randn((B, d.Nk, C), generator=g).to(f16).to(dev)
    v = torch.randn((B, d.Nk, C), generator=g).to(f16).to(dev)
    do = torch.randn((B, d.Nq, C), generator=g).to(f16).to(dev)
    R = d.kH + d.kW + d.kT
    rq = (torch.randn((B * d.Nq * heads, R), generator=g) * 0.1).to(dev)
    oh = tokens.attn_onehot(d, dev)
    scale = D ** -0.5
    o, lse = tokens.attn_fwd(d, q, k, v, scale, rq, True, onehot=oh)
    t_f = timed(lambda: tokens.attn_fwd(d, q, k, v, scale, rq, True, onehot=oh), iters)
    t_b = timed(lambda: tokens.attn_bwd(d, q, k, v, scale, rq, True, o, do, lse, onehot=oh), iters)
    fl = 4.0 * B * heads * d.Nq * d.Nk * D
    print(f"attn B{B} h{heads} D{D} Nq{d.Nq} Nk{d.Nk}: fwd {t_f:7.1f} us {fl / t_f * 1e-6:6.1f} TF | bwd {t_b:7.1f} us "
          f"{2 * fl / t_b * 1e-6:6.1f} TF (algorithmic 8 NqNkD)")


def dw_case(B, heads, Cw, thw, stride, iters, dev):
    f16 = lib.act_dtype()
    C = heads * Cw
    geom = tokens.DwGeom(B, C, Cw, thw, (3, 3, 3), stride, (1, 1, 1), 1)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((geom.rows_in, C), generator=g).to(f16).to(dev)
    w = torch.randn((Cw, 1, 3, 3, 3), generator=g).to(dev)
    dy = torch.randn((geom.rows_out, C), generator=g).to(f16).to(dev)
    dw = torch.empty_like(w)
    t_f = timed(lambda: tokens.dwconv_fwd(x, w, geom), iters)
    t_d = timed(lambda: tokens.dwconv_dgrad(dy, w, geom), iters)
    t_w = timed(lambda: tokens.dwconv_wgrad(x, dy, geom, dw), iters)
    by = 2.0 * C * (geom.rows_in + geom.rows_out)
    print(f"dwconv B{B} C{C} thw{thw} s{stride}: fwd {t_f:6.1f} us {by / t_f * 1e-3:5.0f} GB/s | dgrad {t_d:6.1f} us "
          f"{by / t_d * 1e-3:5.0f} GB/s | wgrad {t_w:6.1f} us {(by + 2.0 * C * geom.rows_out) / t_w * 1e-3:5.0f} GB/s")


def ln_case(M, C, iters, dev):
    """LayerNorm forward / backward (with the residual operand and the two extra column sums, as MultiScaleBlock runs it); the
    operands are cycled over enough copies to come from HBM, not from the 256 MB Infinity Cache."""
    f16 = lib.act_dtype()
    ncopy = max(1, int(1.2e9 // (2.0 * M * C * 4)))
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn((M, C), generator=g).to(f16).to(dev) for _ in range(ncopy)]
    dys = [torch.randn((M, C), generator=g).to(f16).to(dev) for _ in range(ncopy)]
    rs = [torch.randn((M, C), generator=g).to(f16).to(dev) for _ in range(ncopy)]
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dg, db, s0, s1 = (torch.empty(C, device=dev) for _ in range(4))
    y, mean, rstd = tokens.layernorm_fwd(xs[0], gamma, beta, 1e-6)
    out, dx = torch.empty_like(xs[0]), torch.empty_like(xs[0])
    it = [0]

    def fwd():
        it[0] += 1
        tokens.layernorm_fwd(xs[it[0] % ncopy], gamma, beta, 1e-6, out=out)

    def bwd(sums):
        it[0] += 1
        i = it[0] % ncopy
        tokens.layernorm_bwd(dys[i], xs[i], gamma, mean, rstd, dg, db, resid=rs[i], out=dx,
                             sums=((s0, False), (s1, False)) if sums else None)
    t_f = timed(fwd, iters)
    t_b = timed(lambda: bwd(False), iters)
    t_s = timed(lambda: bwd(True), iters)
    by = 2.0 * M * C
    print(f"layernorm M{M} C{C}: fwd {t_f:7.1f} us {2 * by / t_f * 1e-3:6.0f} GB/s | bwd+resid {t_b:7.1f} us {4 * by / t_b * 1e-3:6.0f} GB/s"
          f" | bwd+resid+sums {t_s:7.1f} us {4 * by / t_s * 1e-3:6.0f} GB/s (incl. the finalize launches)")


def colsum_case(M, C, iters, dev):
    """Bias gradient of a Linear layer: column sums of its output gradient (sf_colsum + finalize), operands from HBM."""
    f16 = lib.act_dtype()
    ncopy = max(1, int(1.2e9 // (2.0 * M * C)))
    g = torch.Generator().manual_seed(0)
    base = torch.randn((M, C), generator=g).to(f16).to(dev)
    xs = [base] + [base.clone() for _ in range(ncopy - 1)]
    db = torch.empty(C, device=dev)
    it = [0]

    def run():
        it[0] += 1
        tokens.bias_grad(xs[it[0] % ncopy], db)
    t = timed(run, iters)
    print(f"colsum M{M} C{C}: {t:7.1f} us {2.0 * M * C / t * 1e-3:6.0f} GB/s (incl. the finalize launch)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.only in ("", "attn", "stage3", "stage3attn"):
        attn_case(32, 4, 96, (8, 14, 14), (8, 7, 7), a.iters, dev)      # stage 3 (11 blocks)
    if a.only in ("", "attn"):
        attn_case(32, 1, 96, (8, 56, 56), (8, 7, 7), a.iters, dev)      # block 0
        attn_case(32, 8, 96, (8, 7, 7), (8, 7, 7), a.iters, dev)        # stage 4
    if a.only in ("", "colsum"):
        colsum_case(32 * 25089, 288, a.iters, dev)                       # block 0 d(qkv)
        colsum_case(32 * 1569, 1152, a.iters, dev)                       # stage 3 d(qkv)
    if a.only in ("", "ln"):
        ln_case(32 * 25089, 96, a.iters, dev)                            # block 0
        ln_case(32 * 6273, 192, a.iters, dev)                            # stage 2
        ln_case(32 * 1569, 384, a.iters, dev)                            # stage 3 (11 blocks)
        ln_case(32 * 393, 768, a.iters, dev)                             # stage 4
    if a.only in ("", "dw", "stage3"):
        dw_case(32, 4, 96, (8, 14, 14), (1, 1, 1), a.iters, dev)        # stage 3 q pool
        dw_case(32, 4, 96, (8, 14, 14), (1, 2, 2), a.iters, dev)        # stage 3 k / v pool
    if a.only in ("", "dw"):
        dw_case(32, 1, 96, (8, 56, 56), (1, 1, 1), a.iters, dev)        # block 0 q pool
        dw_case(32, 2, 96, (8, 28, 28), (1, 2, 2), a.iters, dev)        # block 3 q pool (stride 2)


if __name__ == "__main__":
    main()
