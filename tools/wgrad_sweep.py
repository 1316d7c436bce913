"""Weight-gradient schedule sweep on the SlowFast-8x8-R50 layer geometries (round 3): sf_conv_wgrad per layer under
SF_WGRAD2_BLOCKS = target workgroup count (-> number of split partials).  (The first sweep, profiles/r3/r3_v2_wgrad_sweep.md, also
had a six-stage one-workgroup-per-CU ring, SF_WGRAD2_NST=6; it lost everywhere and is gone.)  HIP events around `iters` back-to-back calls; the operands of a layer are
re-created per layer (so small layers are cache-warm, as in tools/microbench.py).
    python tools/wgrad_sweep.py --md gpurun_out/x/wgrad_sweep.md"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slowfast_amd import ops  # noqa: E402
from tools.microbench import LAYERS, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--md", default="")
    ap.add_argument("--filter", default="slow")
    ap.add_argument("--dual", action="store_true", help="compare one / two splits per workgroup (SF_WGRAD2_DUAL) instead of split targets only")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    # (co-tile rows of sf_wgrad2_kernel, workgroup target).  profiles/r3/r3_final_wgrad_sweep.md also has 64-row co-tiles for the wide
    # layers (a knob that existed for that sweep: they lose 20-40 % on res3-res5 and the knob is gone)
    # round 6: (splits per workgroup, workgroup-half target): "1" = sf_wgrad2_kernel<., false>, "2" = the dual kernel (SF_WGRAD2_DUAL)
    variants = [("1", 512), ("2", 384), ("2", 448), ("2", 512), ("2", 640)] if a.dual else [("1", b) for b in (256, 384, 448, 512, 640, 768, 1024)]
    lines = ["| layer | x | " + " | ".join(f"spw{n} b{b}" for n, b in variants) + " | best |", "|---|---:|" + "---:|" * (len(variants) + 1)]
    tot = [0.0] * len(variants)
    best_tot = 0.0
    for name, Ci, T, H, W, Co, k, s, p, cnt in LAYERS:
        if a.filter not in name or "stem" in name:
            continue
        geom = ops.ConvGeom((a.batch, Ci, T, H, W), Co, k, s, p, Cw=Ci)
        x = ops.cl_empty(geom.in_shape, dev)
        x.normal_()
        dy = ops.cl_empty(geom.out_shape, dev)
        dy.normal_()
        dw = torch.empty((Co, Ci) + k, device=dev)
        ts = []
        for bmw, blocks in variants:
            os.environ["SF_WGRAD2_BLOCKS"] = str(blocks)
            os.environ["SF_WGRAD2_DUAL"] = "1" if bmw == "2" else "0"
            geom.ws_bytes = None
            ts.append(timeit(lambda: ops.conv_wgrad(x, dy, geom, dw), a.iters) * 1e3)
        for i, t in enumerate(ts):
            tot[i] += cnt * t
        best_tot += cnt * min(ts)
        b = min(range(len(ts)), key=lambda i: ts[i])
        lines.append(f"| {name} | {cnt} | " + " | ".join(f"{t:.0f}" for t in ts) + f" | spw{variants[b][0]} b{variants[b][1]} |")
        print(lines[-1], flush=True)
    lines.append("| **weighted total (us / step)** | | " + " | ".join(f"{t:.0f}" for t in tot) + f" | {best_tot:.0f} |")
    print(lines[-1])
    if a.md:
        os.makedirs(os.path.dirname(a.md) or ".", exist_ok=True)
        with open(a.md, "w") as f:
            f.write("# sf_conv_wgrad per layer (us per call) vs split target and co-tile height, SlowFast-8x8-R50 geometries, batch %d\n\n" % a.batch)
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
