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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "autocast_yardstick.json"))
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--dtype", default="float16")
    ap.add_argument("--loss-scale", type=float, default=1024.0)
    a = ap.parse_args()
    names = a.cases or sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(mc.GOLDEN_DIR, "*.json"))
                              if not os.path.basename(p).startswith(("eval_", "autocast_")))
    out = {"_meta": {"torch": torch.__version__, "device": a.device, "dtype": a.dtype, "loss_scale": a.loss_scale,
                     "device_name": torch.cuda.get_device_name(0) if a.device.startswith("cuda") else "cpu",
                     "what": "deviation of the pinned oracle graph under torch.autocast from its fp32 CPU run"}}
    if os.path.exists(a.out) and a.cases:
        out.update(json.load(open(a.out)))
    for name in names:
        t = time.time()
        try:
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
