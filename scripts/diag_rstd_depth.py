"""Diagnostic: embedding cosine of the Bottleneck-ResNet executor against the emulating oracle for growing depth, with the
fp32-vs-fp64 band of the oracle itself beside it (is a low cosine noise or wiring?)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import backbones_ref as bb
from tests.golden import common


def cos(a, b):
    return float(torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=1).min())


def run(layers, hw, B, D=64, seed=3, bn3_gamma=1.0):
    from vlsfr_amd.model.resnet_std import ResNet
    sd0 = common.fill_state(bb.resnet_std_state(layers, D, None, hw), seed)
    for k in sd0:
        if k.endswith("bn3.weight"):
            sd0[k] = sd0[k] * bn3_gamma
    net = ResNet(list(layers), feat_dim=D, image_size=hw)
    net.load_state_dict(sd0)
    net = net.cuda().train()
    rng = np.random.default_rng(seed)
    x = common.images_from_u8(common.synth_images_u8(rng, B, hw=hw))
    with torch.no_grad():
        got = net(x.cuda()).cpu()
        outs = {}
        for dt in (torch.float64, torch.float32):
            sd = {k: (v.to(dt).clone() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
            outs[dt] = bb.resnet_std_forward(sd, x.to(dt), layers, True)
        sd = {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        plain = bb.resnet_std_forward(sd, x.double(), layers, False)
    print("bn3 gamma x %.2f layers %s hw %d B %d: gpu vs emu64 %.6f | emu32 vs emu64 %.6f | emu64 vs plain64 %.6f" %
          (bn3_gamma, layers, hw, B, cos(got, outs[torch.float64]), cos(outs[torch.float32], outs[torch.float64]), cos(outs[torch.float64], plain)), flush=True)


if __name__ == "__main__":
    torch.set_num_threads(16)
    run((1, 1, 1, 1), 64, 8)
    run((3, 4, 6, 3), 224, 4)
    for g in (0.5, 0.25, 0.1):
        run((3, 4, 6, 3), 224, 4, bn3_gamma=g)
    run((3, 4, 6, 3), 224, 8, bn3_gamma=0.25)
