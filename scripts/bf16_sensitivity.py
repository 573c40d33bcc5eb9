"""Calibration (CPU, no GPU): how strongly the backbones amplify round-off.  The fp32-vs-fp64
deviation of the oracle's embeddings times 2^15 (the bf16 / fp32 unit-round-off ratio) predicts the
deviation a bf16-storage pipeline must be expected to show against the float64 golden vectors:
MobileFaceNet ~9 %, 4-block iResNet ~3 % at batch 16 (measured 2026-10-04) — the GPU path shows
3-6 % / 1-4 %, i.e. the step-level tolerances in tests/test_step_gpu.py are round-off, not slack."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import backbones_ref as bb
from tests.golden import common

for net, layers in (("mobile", None), ("irtiny", (1, 1, 1, 1))):
    sd0, fwd = bb.make_backbone(net, 64, layers=layers)
    sd = common.fill_state(sd0, 77)
    rng = np.random.default_rng(9)
    x = common.images_from_u8(common.synth_images_u8(rng, 16))
    e64 = fwd({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, x.double())
    e32 = fwd({k: v.clone() for k, v in sd.items()}, x)
    d = (e32.double() - e64).norm(dim=1) / e64.norm(dim=1)
    print("%s: fp32 vs fp64 embedding deviation max %.3e -> predicted bf16 deviation %.3f" % (net, d.max(), d.max() * 32768))
