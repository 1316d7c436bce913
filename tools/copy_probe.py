"""Where do the device-to-device copies (and other non-libsfamd launches) of a training step come from?

One eager step (forward + loss + backward + FlatOptimizer) of a preset under torch.profiler with Python stacks; prints every ATen
operator that launches a device kernel / memcpy, grouped by the innermost slowfast_amd frame of its stack, with the launch count.
Usage (GPU box):  python tools/copy_probe.py --preset SLOWFAST_8x8_R50 --batch 8
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="SLOWFAST_8x8_R50")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    import slowfast_amd as sa
    from slowfast_amd.data_parallel import GradReducer
    from slowfast_amd.optim import construct_optimizer
    import bench

    dev = torch.device("cuda:0")
    cfg = sa.get_preset(a.preset, ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", a.batch] + bench.PRESET_OPTS.get(a.preset, []))
    torch.manual_seed(0)
    model = sa.build_model(cfg, gpu_id=0).train()
    reducer = GradReducer(model, bucket_mb=48.0)
    reducer.attach_torch_param_hooks(model.head.parameters())
    opt = construct_optimizer(model, cfg, reducer, loss_scale=1024.0, dynamic_loss_scale=True)
    T, S = cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE
    fast = torch.randn((a.batch, 3, T, S, S), device=dev)
    if len(cfg.DATA.INPUT_CHANNEL_NUM) == 2:
        idx = torch.linspace(0, T - 1, T // cfg.SLOWFAST.ALPHA).long().to(dev)
        inputs = [torch.index_select(fast, 2, idx).contiguous(), fast]
    else:
        inputs = [fast]
    labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (a.batch,), device=dev)
    loss_fn = bench.make_loss(cfg)

    def step():
        reducer.zero_grad()
        loss = loss_fn(model(inputs).float(), labels)
        (loss * opt.loss_scale).backward()
        opt.finish_and_step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    groups = collections.Counter()
    kern = collections.Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            if "sf_" not in ev.name[:16]:
                kern[ev.name[:70]] += 1
            continue
        if not ev.name.startswith("aten::") or not ev.kernels:
            continue
        if ev.cpu_children and any(c.kernels for c in ev.cpu_children):
            continue                        # count the innermost operator that owns the launch
        frame = "?"
        for fr in ev.stack or []:
            if "slowfast_amd" in fr or "bench.py" in fr or "copy_probe" in fr:
                frame = fr.split("/")[-1][:90]
                break
        groups[(ev.name, ",".join(sorted(set(k.name[:40] for k in ev.kernels))), frame)] += len(ev.kernels)
    print("non-libsfamd device activities of one step:")
    for k, n in kern.most_common(a.top):
        print(f"  {n:5d}  {k}")
    print("ATen operators that launch them, by innermost slowfast_amd frame:")
    for (name, ks, frame), n in groups.most_common(a.top):
        print(f"  {n:5d}  {name:28s} {ks:42s} {frame}")


if __name__ == "__main__":
    main()
