#!/usr/bin/env python3
"""TIMING PROBE: does a BatchNorm-free model (MViTv2-S: samples are independent in forward and backward) run faster as TWO
half-batches on two HIP streams than as one batch on one stream?  (The two pathways of SlowFast gained 7.7 % from exactly that
kind of co-scheduling, engine.run_pathways.)  Eager launches, parameter gradients of both halves land in the same buffers
unsynchronised -- results are garbage, only the time per step is read.
    python tools/batch_split_probe.py [--preset MVITv2_S_16x4] [--batch 32] [--steps 6]"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slowfast_amd as sa  # noqa: E402
from bench import PRESET_OPTS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="MVITv2_S_16x4")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--splits", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = sa.get_preset(a.preset, ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", a.batch] + PRESET_OPTS.get(a.preset, []))
    torch.manual_seed(0)
    model = sa.build_model(cfg, gpu_id=0).train()
    T, S = cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE
    x = torch.randn((a.batch, 3, T, S, S), device=dev)
    y = torch.randint(0, cfg.MODEL.NUM_CLASSES, (a.batch,), device=dev)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)

    def run(nsplit, steps):
        streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(nsplit - 1)]
        xs, ys = x.chunk(nsplit), y.chunk(nsplit)
        main = streams[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for s in streams[1:]:
                s.wait_stream(main)
            for h, s in enumerate(streams):
                with torch.cuda.stream(s):
                    loss = F.cross_entropy(model([xs[h]]).float(), ys[h])
                    (loss * 1024.0).backward()
            for s in streams[1:]:
                main.wait_stream(s)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    def run_graph(nsplit, steps):
        """the same work captured ONCE into a HIP graph (the halves are branches) and replayed: the eager loop enqueues the
        second half only after the first one's ~1100 launches, by which time the GPU has nearly finished it"""
        streams = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
        xs, ys = x.chunk(nsplit), y.chunk(nsplit)
        cap = torch.cuda.Stream(device=dev)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g, stream=cap):
                for s in streams:
                    s.wait_stream(cap)
                for h, s in enumerate(streams):
                    with torch.cuda.stream(s):
                        loss = F.cross_entropy(model([xs[h]]).float(), ys[h])
                        (loss * 1024.0).backward()
                for s in streams:
                    cap.wait_stream(s)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    for nsplit in (1, a.splits, 1, a.splits):
        run(nsplit, 2)
        print(f"{a.preset} batch {a.batch}: {nsplit} stream(s) x batch {a.batch // nsplit}: {run(nsplit, a.steps):.2f} ms per fwd+bwd", flush=True)
    for nsplit in (1, a.splits, 1, a.splits):
        print(f"{a.preset} batch {a.batch}: GRAPH {nsplit} branch(es) x batch {a.batch // nsplit}: {run_graph(nsplit, a.steps):.2f} ms per fwd+bwd", flush=True)


if __name__ == "__main__":
    main()
