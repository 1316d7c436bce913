#!/usr/bin/env python3
"""Throughput of the eval / multi-view test path (tools/test_net.py loop body) on one MI355X: eval-mode forward on
DATA.TEST_CROP_SIZE clips through inference.TestStep (HIP-graph replay), with and without the inference fusion
(BatchNorm folded into the convolutions).  Not the headline metric (bench.py measures training); prints one JSON line.

    python tools/bench_eval.py --preset SLOWFAST_8x8_R50 --batch 32 --steps 10
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="SLOWFAST_8x8_R50")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    import slowfast_amd as sa
    from slowfast_amd import inference
    from slowfast_amd.lib import get_lib
    assert get_lib().backend == "gfx950"
    dev = torch.device("cuda", 0)
    cfg = sa.get_preset(a.preset, ["NUM_GPUS", 1, "MODEL.DROPOUT_RATE", 0.0])
    torch.manual_seed(cfg.RNG_SEED)
    model = sa.build_model(cfg, gpu_id=0)
    T, S = cfg.DATA.NUM_FRAMES, cfg.DATA.TEST_CROP_SIZE
    fast = torch.randn((a.batch, 3, T, S, S), device=dev)
    if len(cfg.DATA.INPUT_CHANNEL_NUM) == 2:
        idx = torch.linspace(0, T - 1, T // cfg.SLOWFAST.ALPHA).long().to(dev)
        inputs = [torch.index_select(fast, 2, idx).contiguous(), fast]
    else:
        inputs = [fast]
    labels = torch.zeros((a.batch,), dtype=torch.long, device=dev)
    ids = torch.arange(a.batch, device=dev)
    out = {"metric": f"clips/sec (eval forward), {a.preset} {T}x{S}^2 synthetic clips, batch {a.batch}", "unit": "clips/s"}
    for fused in (False, True):
        (inference.fuse_for_inference if fused else inference.unfuse)(model.eval())
        step = inference.TestStep(model, num_videos=a.batch, num_clips=1, num_cls=cfg.MODEL.NUM_CLASSES, warmup=1)
        for _ in range(max(a.warmup, 2)):
            step.step(inputs, labels, ids)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step.step(inputs, labels, ids)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["fused" if fused else "running_stats"] = {"clips_per_s": round(a.batch * a.steps / dt, 1),
                                                      "ms_per_step": round(dt / a.steps * 1e3, 3)}
    from slowfast_amd.profiler import KernelProfiler
    with KernelProfiler() as prof, torch.no_grad():          # one eager fused forward with HIP events per launch
        model(inputs)
    summ = prof.summary()
    tot = sum(v["ms"] for v in summ.values())
    out["fused_kernels"] = {k: {"calls": v["calls"], "ms": round(v["ms"], 3), "share": round(v["ms"] / tot, 3),
                                "GB/s": round(v["gbs"], 1), "TFLOP/s": round(v["tflops"], 1)}
                            for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:6]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
