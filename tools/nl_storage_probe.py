#!/usr/bin/env python3
"""CPU probe: how much of the fp16-storage deviation of SlowFast-R101 + Nonlocal (dot_product) gradients comes
from rounding the affinity A = theta^T phi / P (and dA) to 16 bits?  Runs the oracle's storage model on a golden case twice --
A as one 16-bit tensor / A kept exact (what an fp16 hi + lo pair would store) -- and prints the global relative gradient deviation
from the fp32 oracle, overall and for the Nonlocal parameters alone.  Every pass runs its backward through the fp32 pass's ReLU masks / max-pool routes (video_ref.recording_masks).
    python tools/nl_storage_probe.py [r101nl_wc | --full]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import video_ref  # noqa: E402
from tests import model_checks as mc  # noqa: E402


def rel(a, b, keys):
    num = sum(float(((a[k] - b[k]).double() ** 2).sum()) for k in keys)
    den = sum(float((b[k].double() ** 2).sum()) for k in keys)
    return (num / den) ** 0.5


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "r101nl_wc"
    scale = 1024.0
    if name == "--full":            # the SLOWFAST_32x2_R101_50_50@full case of tests/test_model_gpu.py (batch 2, 256^2, 3 boxes per clip)
        preset = "SLOWFAST_32x2_R101_50_50"
        cfg, model, fam, sd, inputs, labels, kw = mc.full_size_case(preset, **mc.FULL_SIZE[preset])
        scale = 64.0
    else:
        gold = mc.load_golden(name)
        cfg = mc.cfg_for(gold)
        fam = mc.family(cfg)
        model, sd, inputs, labels, _, _, ref, _ = mc.oracle_run(gold, cfg)
        kw = {"bboxes": inputs.bboxes} if hasattr(inputs, "bboxes") else {}
    with video_ref.recording_masks() as rec:            # the fp32 pass's own masks / routes, handed to every pass below
        _, _, ref, _ = fam.loss_and_grads(sd, cfg, list(inputs), labels, **kw)
    keys = list(ref.keys())
    nl = [k for k in keys if "nonlocal" in k.lower() or ".nl" in k.lower()]
    for flag in (True, False):
        video_ref.NL_AFFINITY_16BIT = flag
        with video_ref.fp16_storage_model(), video_ref.handed_masks(rec.table):
            _, _, g, _ = fam.loss_and_grads(sd, cfg, list(inputs), labels, loss_scale=scale, **kw)
        print(f"A 16-bit={flag}: grad_global all {rel(g, ref, keys):.5f}  nonlocal params ({len(nl)}) {rel(g, ref, nl) if nl else float('nan'):.5f}", flush=True)
    video_ref.NL_AFFINITY_16BIT = True
    for names in (("theta", "phi"), ("g",), ("y",), ("out",), ("theta", "phi", "g", "y", "out")):
        video_ref.NL_EXACT = frozenset(names)
        with video_ref.fp16_storage_model(), video_ref.handed_masks(rec.table):
            _, _, g, _ = fam.loss_and_grads(sd, cfg, list(inputs), labels, loss_scale=scale, **kw)
        print(f"exact {'+'.join(names)} (A 16-bit): grad_global all {rel(g, ref, keys):.5f}  nonlocal params {rel(g, ref, nl) if nl else float('nan'):.5f}", flush=True)
    video_ref.NL_EXACT = frozenset()


if __name__ == "__main__":
    main()
