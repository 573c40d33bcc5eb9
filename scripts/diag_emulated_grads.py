"""Per-tensor gradient deviation of the GPU step from the float64 oracle, with and without the oracle's bf16-storage
emulation (oracle/backbones_ref.py): shows whether the remaining deviation is a smooth amplification of summation
noise (grows from the head towards the stem) or a rounding point the emulation misses (a jump at one layer type).
    python scripts/diag_emulated_grads.py irtiny 64 512 32"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import backbones_ref as bb  # noqa: E402
from tests import test_step_gpu as T  # noqa: E402

net, D, Q, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
layers = (1, 1, 1, 1) if net == "irtiny" else None
res = {}
for emu in (True, False):
    m, o, x, y, xl, yl = T._emulated_pair(net, D, Q, B, "Arc", 0.5, 31, layers)
    if not emu:
        from oracle import ffc_ref
        o2 = ffc_ref.FFCRef(net, D, Q, 32.0, "Arc", 0.5, 0.99, layers=layers, dtype=torch.float64, emulate_bf16=False)
        o2.probe, o2.gallery, o2.queue = o.probe, o.gallery, o.queue
        o = o2
    loss = m(x.cuda(), y.cuda(), xl, yl)
    loss.backward()
    want = o.forward(x.double(), y.double(), xl, yl)
    want.backward()
    pn = dict(m.probe_net.named_parameters())
    res[emu] = (float(loss.detach()), float(want.detach()),
                {k: (T.rel_l2(pn[k].grad.detach().double().cpu().numpy(), v.grad.numpy()), float(v.grad.norm()))
                 for k, v in o.probe.items() if bb.trainable(k)})
print("loss gpu %.6f  oracle(emulated) %.6f  oracle(plain) %.6f" % (res[True][0], res[True][1], res[False][1]))
for k in res[True][2]:
    print("%-36s |g| %.3e   emulated %.2e   plain %.2e" % (k, res[True][2][k][1], res[True][2][k][0], res[False][2][k][0]))
