"""Reference-derived parity yardstick: what the reference's OWN mixed-precision training path costs on every golden case.

For each training case under tests/golden/ the pinned oracle graph (oracle/video_ref.py, oracle/mvit_ref.py: the torch ops the
reference's modules dispatch to, verified against the unmodified reference by oracle/make_golden.py) is run twice on the same
parameters and clips:
  * fp32 on the CPU (the parity reference), and
  * on the GPU under ``torch.autocast(dtype=float16)`` with a fixed loss scale -- PyTorch-ROCm's MIOpen / rocBLAS kernels, i.e. the
    reference with TRAIN.MIXED_PRECISION True (tools/train_net.py:113 autocast, :152-172 GradScaler).
The deviation of the second from the first (logits, loss, grad-norm, global / per-parameter gradient error, running statistics) is
written to tests/golden/autocast_yardstick.json.  tests/model_checks.py bounds the HIP engine's own deviation from the fp32 oracle
by ``max(north-star tolerance, 1.5 x this)`` -- a number produced by the reference's stack, not by a model of our storage format.

    python tools/autocast_yardstick.py [--out tests/golden/autocast_yardstick.json] [case ...]      (on the GPU box)
"""
import argparse
import glob
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import video_ref  # noqa: E402
from tests import model_checks as mc  # noqa: E402


def deviation(logits, loss, grads, stats, o_logits, o_loss, o_grads, o_stats):
    ogn = float(video_ref.grad_norm(o_grads))
    worst, worst_k = mc._param_worst(grads, o_grads, ogn)
    return {
        "logits": float((logits - o_logits).abs().max() / o_logits.abs().max()),
        "loss": abs(float(loss) - float(o_loss)) / max(1.0, abs(float(o_loss))),
        "grad_norm": abs(float(video_ref.grad_norm(grads)) - ogn) / ogn,
        "grad_global": mc._global_rel(grads, o_grads),
        "param_grad_worst": worst, "param_grad_worst_name": worst_k,
        "running_stats": max([float((stats[k] - v).abs().max() / (v.abs().max() + 1e-6)) for k, v in o_stats.items()
                              if k in stats] + [0.0]),
        "finite": bool(all(torch.isfinite(g).all() for g in grads.values()) and torch.isfinite(logits).all()),
    }


def run_case(name, device, dtype, loss_scale):
    gold = mc.load_golden(name)
    cfg = mc.cfg_for(gold)
    _, sd, inputs, labels, o_logits, o_loss, o_grads, o_stats = mc.oracle_run(gold, cfg)
    fam = mc.family(cfg)
    kw = dict(device=device, autocast_dtype=dtype, loss_scale=loss_scale)
    if isinstance(inputs, mc._WithBoxes):
        kw["bboxes"] = inputs.bboxes
    while True:         # GradScaler semantics: an overflowing step is skipped and the scale halved (train_net.py:152-172)
        logits, loss, grads, stats = fam.loss_and_grads(sd, cfg, list(inputs), labels, **kw)
        if all(torch.isfinite(g).all() for g in grads.values()) or kw["loss_scale"] <= 1.0:
            break
        kw["loss_scale"] /= 2.0
    rec = deviation(logits, loss, grads, stats, o_logits, o_loss, o_grads, o_stats)
    rec["loss_scale_used"] = kw["loss_scale"]
    rec["storage_model"] = {k: v for k, v in mc.storage_model_yardstick(
        name, sd, cfg, inputs, labels, o_logits, o_loss, o_grads, o_stats).items()}
    return rec


def run_eval_case(name, device, dtype):
    """Eval / multi-view test path (tools/test_net.py): the oracle's eval forward under autocast vs its fp32 run."""
    from oracle.make_golden import eval_forward
    import slowfast_amd as sa
    gold = mc.load_golden(name)
    cfg = mc.cfg_for(gold)
    model = sa.MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    fam = mc.family(cfg)
    sd = fam.randomize_state(shapes, gold["param_seed"])
    if "final_bn_gamma_scale" in gold.get("state_tweaks", {}):
        video_ref.scale_final_bn(sd, gold["state_tweaks"]["final_bn_gamma_scale"])
    inputs, _ = video_ref.synthetic_batch(cfg, gold["batch"], gold["data_seed"], crop=gold["test_crop"])
    if fam is video_ref:
        sd = video_ref.calibrate_running_stats(sd, cfg, inputs)
    with torch.no_grad():
        o_probs = eval_forward(sd, cfg, inputs)
        sdd = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in sd.items()}
        with torch.autocast(torch.device(device).type, dtype=dtype):
            probs = eval_forward(sdd, cfg, [x.to(device) for x in inputs]).float().cpu()
        with video_ref.fp16_storage_model():
            sm = eval_forward(sd, cfg, inputs)
    pmax = float(o_probs.max())
    return {"probs": float((probs - o_probs).abs().max() / pmax), "finite": bool(torch.isfinite(probs).all()),
            "storage_model": {"probs": float((sm - o_probs).abs().max() / pmax)}}


def run_full_case(key, device, dtype, loss_scale):
    """Full-size cases of tests/model_checks.py ("<preset>@full": BASELINE configs at batch 2; "<preset>@b32": the benchmark's batch
    without any conditioning device): the SAME parameters and clips the GPU tests use, the oracle graph under autocast on the GPU
    against its fp32 CPU run."""
    preset, kind = key.split("@")
    spec = mc.FULL_SIZE[preset] if kind == "full" else mc.BATCH32[preset]
    cfg, model, fam, sd, inputs, labels, kw = mc.full_size_case(preset, **spec)
    del model
    o_logits, o_loss, o_grads, o_stats = fam.loss_and_grads(sd, cfg, list(inputs), labels, **kw)
    kw = dict(kw, device=device, autocast_dtype=dtype, loss_scale=loss_scale)
    while True:
        logits, loss, grads, stats = fam.loss_and_grads(sd, cfg, list(inputs), labels, **kw)
        if all(torch.isfinite(g).all() for g in grads.values()) or kw["loss_scale"] <= 1.0:
            break
        kw["loss_scale"] /= 2.0
    rec = deviation(logits, loss, grads, stats, o_logits, o_loss, o_grads, o_stats)
    rec["logits_l2"] = float((logits - o_logits).norm() / o_logits.norm())
    rec["loss_scale_used"] = kw["loss_scale"]
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*")
    ap.add_argument("--full", action="store_true", help="the named cases are full-size keys (PRESET@full | PRESET@b32); default: all")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "autocast_yardstick.json"))
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--dtype", default="float16")
    ap.add_argument("--loss-scale", type=float, default=1024.0)
    a = ap.parse_args()
    if a.full:
        names = a.cases or [k + "@full" for k in mc.FULL_SIZE] + [k + "@b32" for k in mc.BATCH32]
    else:
        names = a.cases or sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(mc.GOLDEN_DIR, "*.json"))
                                  if not os.path.basename(p).startswith("autocast_"))
    out = {"_meta": {"torch": torch.__version__, "device": a.device, "dtype": a.dtype, "loss_scale": a.loss_scale,
                     "device_name": torch.cuda.get_device_name(0) if a.device.startswith("cuda") else "cpu",
                     "what": "deviation of the pinned oracle graph under torch.autocast from its fp32 CPU run"}}
    if os.path.exists(a.out) and (a.cases or a.full):       # a partial run keeps every other entry (and the first run's _meta)
        out.update(json.load(open(a.out)))
    for name in names:
        t = time.time()
        try:
            if "@" in name:
                out[name] = run_full_case(name, a.device, getattr(torch, a.dtype), a.loss_scale)
            elif name.startswith("eval_"):
                out[name] = run_eval_case(name, a.device, getattr(torch, a.dtype))
            else:
                out[name] = run_case(name, a.device, getattr(torch, a.dtype), a.loss_scale)
            print(name, {k: (round(v, 6) if isinstance(v, float) else v) for k, v in out[name].items() if k != "storage_model"},
                  f"{time.time() - t:.1f}s", flush=True)
        except Exception as e:       # noqa: BLE001 -- a case the stock kernels cannot run is recorded, not fatal
            out[name] = {"error": repr(e)[:300]}
            print(name, "ERROR", repr(e)[:300], flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
