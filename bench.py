#!/usr/bin/env python3
"""Headline benchmark: SlowFast-8x8-R50 training step (forward + backward + SGD) on synthetic Kinetics clips.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU (RANK/LOCAL_RANK/WORLD_SIZE from the env), per-GPU batch 32 (weak scaling), gradients
all-reduced over RCCL by slowfast_amd.data_parallel.GradReducer overlapped with backward.  Rank 0 prints ONE
JSON line.  `value` = clips/s of the whole job with the clips already resident in HBM.  After the timed
region one more step is run with HIP events around every libsfamd launch (slowfast_amd.profiler) to report the
dominant kernel against its roofline, and (N=1 only) the CPU oracle is timed on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_PEAK_TFLOPS = 2500.0    # dense fp16/bf16 MFMA
# per clip, SlowFast-8x8-R50 32x224^2: MAC_fwd 50.309 G -> 6*MAC train flops; boundary elements
# E = 216.5 M -> byte floor 5*E*2 B = 2.165 GB
TRAIN_GFLOP_PER_CLIP = {"SLOWFAST_8x8_R50": 301.9, "C2D_8x8_R50": 117.0, "MVITv2_S_16x4": 383.6, "X3D_M": 28.4,
                        "SLOWFAST_32x2_R101_50_50": 879.4}
BYTE_FLOOR_GB_PER_CLIP = {"SLOWFAST_8x8_R50": 2.165, "C2D_8x8_R50": 1.019, "MVITv2_S_16x4": 1.845, "X3D_M": 0.989,
                          "SLOWFAST_32x2_R101_50_50": 4.729}
METRIC_NAME = {"SLOWFAST_8x8_R50": "SlowFast-8x8-R50 32x224^2", "C2D_8x8_R50": "C2D-R50 8x224^2",
               "MVITv2_S_16x4": "MViTv2-S 16x224^2", "X3D_M": "X3D-M 16x224^2",
               "SLOWFAST_32x2_R101_50_50": "SlowFast-32x2-R101+Nonlocal (AVA RoI head, 3 boxes/clip) 32x256^2"}
# synthetic-run overrides: stochastic ops off so that runs are comparable and parity-checkable;
# BASELINE config 5 is quoted on AVA-shaped 32x256^2 clips (ResNetRoIHead on 3 synthetic boxes per clip)
PRESET_OPTS = {"MVITv2_S_16x4": ["MVIT.DROPPATH_RATE", 0.0, "MODEL.DROPOUT_RATE", 0.0],
               "SLOWFAST_32x2_R101_50_50": ["DATA.TRAIN_CROP_SIZE", 256],
               # the shipped yaml's CLS_EMBED_ON True is rejected by the reference's own constructor (no cls token in rev)
               "REV_MVIT_B_16x4_CONV": ["MVIT.CLS_EMBED_ON", False, "MVIT.DROPPATH_RATE", 0.0, "MODEL.DROPOUT_RATE", 0.0]}
BOXES_PER_CLIP = 3           # synthetic AVA batches: boxes (R, 5) = [batch index, x1, y1, x2, y2] in crop pixels


def make_loss(cfg):
    """slowfast/models/losses.py:61-69: cross_entropy | soft_cross_entropy | bce (nn.BCELoss on the RoI head's sigmoid
    outputs; BCE-with-logits if the detection head is switched off and the basic head returns logits)."""
    if cfg.MODEL.LOSS_FUNC == "bce":
        return F.binary_cross_entropy if cfg.DETECTION.ENABLE else F.binary_cross_entropy_with_logits
    return F.cross_entropy


# libsfamd entry point -> the HIP kernels it launches (for mapping rocprofv3 PMC traffic, collected per kernel in a
# separate --pmc pass and committed under profiles/, onto the entry point the in-process profiler times)
ENTRY_KERNELS = {
    "sf_conv_wgrad": ("sf_wgrad_kernel", "sf_wgrad2_kernel", "sf_wgrad2_rowtab_kernel", "sf_stem_wgrad_kernel",
                      "sf_wgrad_reduce_kernel"),
    "sf_bn_bwd_apply": ("sf_bn_bwd_apply_kernel",), "sf_bn_bwd_reduce": ("sf_bn_bwd_reduce_kernel",),
    "sf_bn_act": ("sf_bn_act_kernel",), "sf_dwconv_fwd": ("sf_dwconv_fwd",), "sf_dwconv_dgrad": ("sf_dwconv_dgrad",),
    "sf_dwconv_wgrad": ("sf_dwconv_wgrad",), "sf_softmax_fwd": ("sf_softmax_fwd_kernel",),
    "sf_softmax_bwd": ("sf_softmax_bwd_kernel",),
    "sf_attn_fwd": ("sf_attn_fwd_kernel",),
    "sf_attn_bwd": ("sf_attn_bwd_dq_kernel", "sf_attn_reduce_kernel", "sf_attn_bwd_dkv_kernel"),
}


def pmc_traffic(entry, preset, launches_per_step):
    """HBM bytes per launch of `entry` from the committed PMC pass of this preset (FETCH_SIZE doubled per
    MI355X_MICROARCH.md, WRITE_SIZE as reported; tools/pmc_traffic.py), or None."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{preset}.json")
    if entry not in ENTRY_KERNELS or not os.path.exists(path):
        return None
    with open(path) as f:
        doc = json.load(f)
    kern = doc["kernels"]
    pmc_traffic.build_id = doc.get("sf_build_id")       # the library build the PMC passes were taken on
    tot, steps = 0.0, None
    for name, v in kern.items():
        if any(name.replace("void ", "").startswith(k) for k in ENTRY_KERNELS[entry]):
            tot += v["hbm_bytes_per_launch_corrected"] * v["launches"]
            if name.replace("void ", "").startswith(ENTRY_KERNELS[entry][-1]):
                steps = (steps or 0) + v["launches"]
    if not tot or not steps:
        return None
    return tot / steps          # per launch of the entry point (its last kernel runs once per call)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--preset", default="SLOWFAST_8x8_R50")
    ap.add_argument("--loss-scale", type=float, default=1024.0)
    ap.add_argument("--bucket-mb", type=float, default=48.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--cpu-baseline-clips", type=int, default=2)
    ap.add_argument("--cpu-baseline-threads", type=int, default=0, help="0 = min(64, os.cpu_count())")
    ap.add_argument("--cpu-baseline-timeout", type=float, default=240.0)
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) time the CPU oracle and exit")
    ap.add_argument("--no-graph", action="store_true", help="issue every kernel from Python instead of a HIP graph")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="NOT a measurement: run the same launch plumbing (rank spawn, process group, reducer, TrainStep, JSON line) on CPU "
                         "through the host functional simulator (tests/hostsim) with gloo and a shrunken model -- checks that "
                         "`bench.py --gpus N` works without a GPU node")
    ap.add_argument("--torch-optimizer", action="store_true",
                    help="torch.optim fused SGD / AdamW + separate unscale / norm / clip passes (round-1 step glue) instead of "
                         "slowfast_amd.optim.FlatOptimizer (one norm pass + one fused update, device-side dynamic loss scale)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the second model of BASELINE.json's metric (MViTv2-S) that the default run appends as `secondary`")
    ap.add_argument("--no-others", action="store_true",
                    help="skip the remaining GPU configurations of BASELINE.json (X3D-M batch 64, SlowFast-R101 + Nonlocal AVA batch 16) "
                         "that the default run appends as `others` (--no-secondary skips them too)")
    ap.add_argument("--master-port", type=int, default=0, help="(self-spawned multi-GPU runs) rendezvous port, 0 = pick a free one")
    return ap.parse_args()


def make_optimizer(model, cfg):
    """SGD as the reference constructs it (slowfast/models/optimizer.py:15-140): BN parameters get
    BN.WEIGHT_DECAY, the rest SOLVER.WEIGHT_DECAY; momentum / nesterov from SOLVER."""
    bn, rest, zero = [], [], []
    for m in model.modules():
        is_bn = isinstance(m, torch.nn.modules.batchnorm._NormBase)
        for p in m.parameters(recurse=False):
            if is_bn:
                bn.append(p)
            elif cfg.SOLVER.ZERO_WD_1D_PARAM and (p.dim() == 1 or p.shape == (1, 1, p.shape[-1])):
                zero.append(p)                      # optimizer.py:57-66: no decay on 1-D parameters (MViT recipe)
            else:
                rest.append(p)
    groups = [g for g in ({"params": bn, "weight_decay": cfg.BN.WEIGHT_DECAY},
                          {"params": rest, "weight_decay": cfg.SOLVER.WEIGHT_DECAY},
                          {"params": zero, "weight_decay": 0.0}) if g["params"]]
    if cfg.SOLVER.OPTIMIZING_METHOD == "adamw":    # slowfast/models/optimizer.py:118-125
        return torch.optim.AdamW(groups, lr=cfg.SOLVER.BASE_LR, betas=(0.9, 0.999), eps=1e-08, fused=True)
    kw = dict(lr=cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM, dampening=cfg.SOLVER.DAMPENING,
              nesterov=cfg.SOLVER.NESTEROV)
    try:
        return torch.optim.SGD(groups, fused=True, **kw)
    except Exception:
        return torch.optim.SGD(groups, foreach=True, **kw)


def cpu_baseline_subprocess(a):
    """Time the CPU oracle in a child process with a hard time limit (the default bench run must finish in minutes
    whatever the host looks like); returns the cpu_baseline object."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--preset", a.preset,
           "--cpu-baseline-clips", str(a.cpu_baseline_clips), "--cpu-baseline-threads", str(a.cpu_baseline_threads)]
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=a.cpu_baseline_timeout, env=env)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "clips/s", "cores": None, "kind": "port",
                "sample": "failed: " + (r.stderr.strip().splitlines() or ["no output"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "clips/s", "cores": None, "kind": "port",
                "sample": f"timed out after {a.cpu_baseline_timeout:.0f} s"}


def cpu_baseline(cfg, clips, threads=0):
    """The CPU oracle (plain torch fp32 restatement of the reference graph) on the host cores."""
    from oracle import mvit_ref, video_ref
    import slowfast_amd as sa
    fam = mvit_ref if cfg.MODEL.MODEL_NAME == "MViT" else video_ref
    torch.set_num_threads(threads or min(64, os.cpu_count() or 1))
    model = sa.MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k in sd:                         # ZERO_INIT_FINAL_BN gammas -> 1 so backward does full work
        if k.endswith("c_bn.weight"):
            sd[k].fill_(1.0)
    inputs, labels = video_ref.synthetic_batch(cfg, clips, seed=0)
    kw = {}
    if cfg.DETECTION.ENABLE:
        kw["bboxes"] = video_ref.synthetic_boxes(cfg, clips, seed=1, per_clip=BOXES_PER_CLIP)
        labels = (torch.rand((kw["bboxes"].shape[0], cfg.MODEL.NUM_CLASSES)) < 0.05).float()
    fam.loss_and_grads(sd, cfg, inputs, labels, **kw)      # warm-up
    best, iters = 1e30, 2
    for _ in range(iters):
        t0 = time.perf_counter()
        fam.loss_and_grads(sd, cfg, inputs, labels, **kw)
        best = min(best, time.perf_counter() - t0)
    return {"value": clips / best, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{clips} clips x (fwd + {cfg.MODEL.LOSS_FUNC} loss + bwd), best of {iters} timed iterations after 1 warm-up, "
                      f"torch {torch.__version__} CPU fp32; 'port' = the oracle restatement of the reference graph (oracle/), "
                      "pinned bit-for-bit to the unmodified reference by tests/golden -- /root/reference does not exist on the GPU box"}


# --dry-run-cpu: SlowFast-R18-like widths on 8 x 32^2 clips (the geometry of tests/golden/slowfast_tiny.json)
DRY_RUN_OPTS = ["NUM_GPUS", 0, "MODEL.DROPOUT_RATE", 0.0, "MODEL.NUM_CLASSES", 10, "DATA.TRAIN_CROP_SIZE", 32,
                "RESNET.WIDTH_PER_GROUP", 16, "RESNET.DEPTH", 18, "DATA.NUM_FRAMES", 8, "SLOWFAST.BETA_INV", 2,
                "RESNET.NUM_BLOCK_TEMP_KERNEL", "[[2, 2], [2, 2], [2, 2], [2, 2]]"]


def spawn_ranks(a):
    """`python bench.py --gpus N` without a torchrun environment: re-execute under torch.distributed.run, one rank per
    GPU on 127.0.0.1 (the contract's own launch line), and pass its exit code through."""
    import socket
    import subprocess
    port = a.master_port
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # The ranks inherit this process's environment unchanged.  Multi-process GPU work on this pool needs
    # HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC; the image exports it): it is NOT set here behind the caller's back
    #  -- only reported when it is missing, so that an RCCL `hipIpcGetMemHandle` failure is explained.
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ and not a.dry_run_cpu:
        print("bench.py: HSA_ENABLE_IPC_MODE_LEGACY is not set; RCCL across processes may need HSA_ENABLE_IPC_MODE_LEGACY=0",
              file=sys.stderr)
    return subprocess.call(cmd, env=dict(os.environ))


def run_preset(a, preset, batch, steps, warmup, rank, local, world, dev, kernel_profile, cpu_base):
    """Builds the model of `preset`, times `steps` training steps, returns the result object (rank 0) or None."""
    import slowfast_amd as sa
    from slowfast_amd import lib as sflib
    from slowfast_amd.data_parallel import GradReducer
    from slowfast_amd.profiler import KernelProfiler

    cfg = sa.get_preset(preset, ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", batch * world] + PRESET_OPTS.get(preset, [])
                        + (DRY_RUN_OPTS if a.dry_run_cpu else []))
    torch.manual_seed(cfg.RNG_SEED)
    model = sa.build_model(cfg, gpu_id=local).train()
    if world > 1:                        # replicas start identical (DDP's initial broadcast)
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
    reducer = GradReducer(model, bucket_mb=a.bucket_mb)
    reducer.attach_torch_param_hooks(model.head.parameters())
    if a.torch_optimizer:
        opt = make_optimizer(model, cfg)
    else:       # tools/train_net.py:150-172 + optimizer.step() as three launches over the flat buffers, GradScaler on the device
        from slowfast_amd.optim import construct_optimizer
        opt = construct_optimizer(model, cfg, reducer, loss_scale=a.loss_scale, dynamic_loss_scale=True)

    g = torch.Generator(device=dev).manual_seed(cfg.RNG_SEED + rank)
    T, S = cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE
    fast = torch.randn((batch, 3, T, S, S), generator=g, device=dev)
    bboxes = None
    if cfg.DETECTION.ENABLE:             # AVA: boxes + one multi-hot action label row per box
        R = batch * BOXES_PER_CLIP
        xy = torch.rand((R, 2), generator=g, device=dev) * 0.6 * S
        wh = (torch.rand((R, 2), generator=g, device=dev) * 0.35 + 0.05) * S
        bidx = torch.arange(batch, device=dev).repeat_interleave(BOXES_PER_CLIP).float()[:, None]
        bboxes = torch.cat([bidx, xy, torch.minimum(xy + wh, torch.full_like(xy, S - 1.0))], 1)
        labels = (torch.rand((R, cfg.MODEL.NUM_CLASSES), generator=g, device=dev) < 0.05).float()
    elif cfg.MODEL.LOSS_FUNC == "bce":
        labels = (torch.rand((batch, cfg.MODEL.NUM_CLASSES), generator=g, device=dev) < 0.05).float()
    else:
        labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (batch,), generator=g, device=dev)
    loss_fn = make_loss(cfg)
    if len(cfg.DATA.INPUT_CHANNEL_NUM) == 2:
        idx = torch.linspace(0, T - 1, T // cfg.SLOWFAST.ALPHA).long().to(dev)
        inputs = [torch.index_select(fast, 2, idx).contiguous(), fast]
    else:
        inputs = [fast]

    step_model = model
    if bboxes is not None:               # the boxes travel as the last element of the (static) input list
        class _WithBoxes(torch.nn.Module):
            def __init__(self, m):
                super().__init__()
                self.m = m

            def forward(self, xs):
                return self.m(xs[:-1], xs[-1])
        step_model = _WithBoxes(model)
        inputs = inputs + [bboxes]

    from slowfast_amd.step import TrainStep
    train_step = TrainStep(step_model, reducer, opt, loss_fn, loss_scale=a.loss_scale,
                           use_graph=not a.no_graph, warmup=1, clip_grad_l2norm=cfg.SOLVER.CLIP_GRAD_L2NORM,
                           clip_grad_val=cfg.SOLVER.CLIP_GRAD_VAL)

    def step():
        return train_step(inputs, labels)

    def eager_step():
        reducer.zero_grad()
        logits = step_model(inputs)
        loss = loss_fn(logits.float(), labels)
        if a.torch_optimizer:
            (loss * a.loss_scale).backward()
            reducer.finish(loss_scale=a.loss_scale)
        else:
            (loss * opt.loss_scale).backward()
            opt.finish_and_step()
            return loss
        opt.step()
        return loss

    def fence():
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    warmup = max(warmup, 0 if a.no_graph else 2)             # graph mode: 1 eager step + the capturing step
    for _ in range(warmup):
        loss = step()
    st = train_step.static_inputs()
    if st is not None:
        # "inputs already resident in HBM": the batch lives in the buffers the captured graph reads (where a loader such as
        # data.pack_pathways_u8 would put it), so the timed steps do not re-copy 0.7 GB of clips device-to-device per iteration
        inputs, labels = st
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.detach())

    # N > 1: what a scaling efficiency below 0.9 would have to be attributed to.  `overlap_log` = collectives
    # still in flight when the LAST backward segment starts (per timed iteration); `one_stream` = the same binary with the
    # pathway / branch streams off (engine.run_pathways / run_branches), i.e. without RCCL's kernels sharing the chip with a
    # two-queue backward.
    multi = None
    # (SF_BENCH_ATTRIB=1 runs the same block on one GPU: the only way to exercise it on a one-GPU box)
    if (world > 1 or os.environ.get("SF_BENCH_ATTRIB") == "1") and not a.dry_run_cpu:
        from slowfast_amd import engine as _eng
        multi = {"overlap_log": list(train_step.overlap_log)[-steps:], "segments": len(getattr(train_step, "_seg_params", []) or []),
                 "buckets": len(reducer.buckets)}
        keep = (_eng.PATHWAY_STREAMS, _eng.BRANCH_STREAMS)
        _eng.PATHWAY_STREAMS = _eng.BRANCH_STREAMS = False
        try:
            ts1 = TrainStep(step_model, reducer, opt, loss_fn, loss_scale=a.loss_scale, use_graph=not a.no_graph, warmup=1,
                            clip_grad_l2norm=cfg.SOLVER.CLIP_GRAD_L2NORM, clip_grad_val=cfg.SOLVER.CLIP_GRAD_VAL)
            for _ in range(3):
                ts1(inputs, labels)
            fence()
            t1 = time.perf_counter()
            for _ in range(steps):
                ts1(inputs, labels)
            fence()
            d1 = time.perf_counter() - t1
            tt = torch.tensor([d1], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            multi["one_stream"] = {"ms_per_step": round(float(tt.item()) / steps * 1e3, 3),
                                   "overlap_log": list(ts1.overlap_log)[-steps:],
                                   "note": "SF_PATHWAY_STREAMS=0 SF_BRANCH_STREAMS=0 equivalent, same process, after the timed region"}
            del ts1
        except Exception as e:      # an attribution aid must never take the measurement down with it (same code on every rank)
            multi["one_stream"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            _eng.PATHWAY_STREAMS, _eng.BRANCH_STREAMS = keep

    kernels, roof = {}, None
    if kernel_profile:
        with KernelProfiler() as prof:
            eager_step()
        summ = prof.summary()
        # the same step on ONE stream (engine.run_pathways / run_branches off): what the dominant kernel does when nothing runs
        # beside it -- reported next to the figure of the step as it is timed (`roofline.serial`), never instead of it
        from slowfast_amd import engine as _eng
        serial = None
        if dev.type == "cuda" and (_eng.PATHWAY_STREAMS or _eng.BRANCH_STREAMS):
            keep = (_eng.PATHWAY_STREAMS, _eng.BRANCH_STREAMS)
            _eng.PATHWAY_STREAMS = _eng.BRANCH_STREAMS = False
            try:
                with KernelProfiler() as prof1:
                    eager_step()
                serial = prof1.summary()
            finally:
                _eng.PATHWAY_STREAMS, _eng.BRANCH_STREAMS = keep
        tot = sum(v["ms"] for v in summ.values())
        for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:8]:
            kernels[k] = {"calls": v["calls"], "ms": round(v["ms"], 3), "avg_ms": round(v["avg_ms"], 4),
                          "share": round(v["ms"] / tot, 3), "GB/s": round(v["gbs"], 1), "TFLOP/s": round(v["tflops"], 1)}
        name, v = max(summ.items(), key=lambda kv: kv[1]["ms"])
        ai = v["flops"] / max(v["bytes"], 1.0)
        if ai > MFMA_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS:
            roof = {"kernel": name, "bound": "mfma", "achieved": round(v["tflops"], 2), "peak": MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(v["tflops"] / MFMA_PEAK_TFLOPS, 4), "traffic": None}
        else:
            roof = {"kernel": name, "bound": "hbm", "achieved": round(v["gbs"], 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(v["gbs"] / HBM_PEAK_GBS, 4), "traffic": None}
        t = pmc_traffic(name, preset, v["calls"])
        if t is not None:
            roof["traffic"] = round(t, 0)
            roof["traffic_note"] = ("HBM bytes per launch from the COMMITTED rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                    f"(profiles/pmc_traffic_{preset}.json, not this run); algorithmic bytes per launch = "
                                    f"{v['bytes'] / max(v['calls'], 1):.0f}")
            roof["traffic_sf_build_id"] = getattr(pmc_traffic, "build_id", None)
            roof["sf_build_id"] = sflib.get_lib().build_id
        roof["avg_launch_ms"] = round(v["avg_ms"], 4)
        roof["launches_per_step"] = v["calls"]
        roof["kernel_ms_per_step"] = round(tot, 2)
        roof["timing_note"] = ("entry points timed with HIP events on the launch stream in ONE extra EAGER step after the timed "
                               "region: includes event / launch overhead; independent pathways / branches run on two streams "
                               "(engine.run_pathways / run_branches), so kernels overlap -- the per-kernel sum exceeds "
                               "ms_per_step and a kernel's duration includes what its neighbour on the other stream costs it; "
                               "the rocprofv3 --kernel-trace table of the same command is under profiles/")
        if serial is not None and name in serial:
            sv = serial[name]
            key, peak = ("tflops", MFMA_PEAK_TFLOPS) if roof["bound"] == "mfma" else ("gbs", HBM_PEAK_GBS)
            roof["serial"] = {"achieved": round(sv[key], 1), "frac": round(sv[key] / peak, 4), "avg_launch_ms": round(sv["avg_ms"], 4),
                              "kernel_ms_per_step": round(sum(x["ms"] for x in serial.values()), 2),
                              "note": "the same eager step on ONE stream: the kernel without a neighbour"}

    out = None
    if rank == 0:
        clips = batch * world * steps
        value = clips / dt
        ms = dt / steps * 1e3
        out = {
            "metric": f"clips/sec (fwd+bwd), {METRIC_NAME.get(preset, preset)} synthetic clips, "
                      f"per-GPU batch {batch}",
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": sflib.ACT_MODE, "data": "DRY RUN on the CPU host simulator with a shrunken model -- not a measurement" if a.dry_run_cpu else "synthetic",
            "config": {"workload": f"{preset}: forward + {cfg.MODEL.LOSS_FUNC} loss + backward + {cfg.SOLVER.OPTIMIZING_METHOD} step, "
                                   f"inputs resident in HBM, "
                                   f"per-GPU batch {batch}", "global_batch": batch * world,
                       "parallelism": f"dp{world}", "loss_scale": a.loss_scale, "bucket_mb": a.bucket_mb,
                       "launch": "eager" if a.no_graph else "hip-graph(fwd+bwd) + eager all-reduce/optimizer",
                       "optimizer": "torch fused + separate unscale/norm/clip" if a.torch_optimizer else
                                    "FlatOptimizer (fused unscale+norm+clip+update, dynamic loss scale on device)"},
            "per_gpu_clips_per_s": round(value / world, 2), "final_loss": round(final_loss, 4),
        }
        if preset in BYTE_FLOOR_GB_PER_CLIP:
            per_gpu = value / world
            out["model_roofline"] = {
                "bound": "hbm", "achieved": round(per_gpu * BYTE_FLOOR_GB_PER_CLIP[preset], 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(per_gpu * BYTE_FLOOR_GB_PER_CLIP[preset] / HBM_PEAK_GBS, 4),
                "note": "whole step vs the fused-ideal byte floor 5*E*2B per clip (SURVEY.md 8d)",
                "mfma_tflops": round(per_gpu * TRAIN_GFLOP_PER_CLIP[preset] / 1e3, 1)}
        if multi is not None:
            out["multi_gpu"] = multi
        if roof is not None:
            out["roofline"] = roof
            out["kernels"] = kernels
        if cpu_base:
            a2 = argparse.Namespace(**vars(a))
            a2.preset = preset
            out["cpu_baseline"] = cpu_baseline_subprocess(a2)
    # release this model before another preset is built in the same process
    reducer.close()
    del train_step, reducer, opt, model, step_model, inputs, fast
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    if a.cpu_baseline_only:
        import slowfast_amd as sa
        cfg = sa.get_preset(a.preset, ["NUM_GPUS", 0, "TRAIN.BATCH_SIZE", a.cpu_baseline_clips]
                            + PRESET_OPTS.get(a.preset, []))
        print(json.dumps(cpu_baseline(cfg, a.cpu_baseline_clips, a.cpu_baseline_threads)), flush=True)
        return
    if a.dry_run_cpu and "SFAMD_LIBRARY" not in os.environ:
        from slowfast_amd import build_ext
        from slowfast_amd import lib as sflib
        os.environ["SFAMD_LIBRARY"] = build_ext.build_hostsim(act=sflib.ACT_MODE)      # inherited by the spawned ranks
        os.environ.setdefault("SF_SIM_THREADS", "2")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    from slowfast_amd.lib import get_lib
    if a.dry_run_cpu:
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group(backend="gloo")
        assert get_lib().backend == "hostsim", "--dry-run-cpu runs on the host simulator (SFAMD_LIBRARY)"
        a.no_graph, a.no_secondary, a.no_kernel_profile, a.no_cpu_baseline = True, True, True, True
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if world > 1:
            dist.init_process_group(backend="nccl", device_id=dev)
        assert get_lib().backend == "gfx950", "bench.py measures the HIP library only"

    out = run_preset(a, a.preset, a.batch, a.steps, a.warmup, rank, local, world, dev,
                     kernel_profile=not a.no_kernel_profile, cpu_base=(world == 1 and not a.no_cpu_baseline))
    # BASELINE.json's metric names two models: the default single-GPU run appends the second one (MViTv2-S 16x224^2, batch
    # 32) measured the same way in the same process, so that the driver's line carries both
    if a.preset == "SLOWFAST_8x8_R50" and world == 1 and not a.no_secondary and a.batch == 32:
        a.cpu_baseline_timeout = min(a.cpu_baseline_timeout, 150.0)     # the second model's CPU sample is bounded harder
        sec = run_preset(a, "MVITv2_S_16x4", 32, min(a.steps, 10), min(a.warmup, 3), rank, local, world, dev,
                         kernel_profile=not a.no_kernel_profile, cpu_base=not a.no_cpu_baseline)
        if out is not None and sec is not None:
            out["secondary"] = {k: sec[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config",
                                                    "model_roofline", "roofline", "kernels", "final_loss", "cpu_baseline")
                                if k in sec}
            out["secondary"]["preset"] = "MVITv2_S_16x4"
        # ... and the other two GPU configurations BASELINE.json names (configs[2]: X3D-M batch 64, configs[4]: SlowFast-R101 +
        # Nonlocal on AVA-shaped clips), so that the driver's record covers every one of them; CPU samples bounded harder still
        if not a.no_others:
            others = []
            for preset, batch in (("X3D_M", 64), ("SLOWFAST_32x2_R101_50_50", 16)):
                a.cpu_baseline_timeout = min(a.cpu_baseline_timeout, 90.0)
                r = run_preset(a, preset, batch, min(a.steps, 10), min(a.warmup, 3), rank, local, world, dev,
                               kernel_profile=not a.no_kernel_profile, cpu_base=not a.no_cpu_baseline)
                if r is not None:
                    o = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config", "model_roofline",
                                           "roofline", "final_loss", "cpu_baseline") if k in r}
                    o["preset"] = preset
                    others.append(o)
            if out is not None:
                out["others"] = others
    if rank == 0 and out is not None:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
