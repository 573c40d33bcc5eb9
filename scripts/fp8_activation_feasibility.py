"""Would e4m3 STORAGE of MobileFaceNet's activations meet SURVEY 8(d)'s fp8 tolerance (embedding cosine >= 0.99, loss 5e-2)?
CPU experiment on the oracle: the bf16-emulating forward with every stored activation (convolution outputs and
BatchNorm / PReLU outputs) additionally rounded to e4m3 after a per-tensor power-of-two scale that puts its maximum
just inside the format's range; embeddings against the plain float64 forward and against the bf16-emulating one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import backbones_ref as bb
from tests.golden import common

def fp8_round(t, which):
    if which == "none":
        return t
    amax = float(t.detach().abs().max())
    if amax == 0:
        return t
    scale = 2.0 ** np.floor(np.log2(448.0 / amax))
    q = (t.detach().float() * scale).to(torch.float8_e4m3fn).float() / scale
    return t + (q.to(t.dtype) - t.detach())

def run(which, B=32, D=512, seed=3):
    sd = common.fill_state(bb.mobilefacenet_state(D), seed)
    sd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    rng = np.random.default_rng(seed)
    x = common.images_from_u8(common.synth_images_u8(rng, B)).double()
    E = bb.Emu(which != "plain")
    if which in ("conv", "act", "all"):
        R0 = E.R
        state = {"k": 0}
        class E8(bb.Emu):
            def R(self, t):            # mobile_unit calls R twice per unit: convolution output, then BatchNorm/PReLU output
                k = state["k"]
                state["k"] += 1
                is_conv = (k % 2 == 0)
                if which == "all" or (which == "conv") == is_conv:
                    return fp8_round(R0(t), "all")
                return R0(t)
        E = E8(True)
    orig = bb.Emu
    bb.Emu = lambda on: E
    try:
        with torch.no_grad():
            out = bb.mobilefacenet_forward(sd, x, emulate_bf16=True)
    finally:
        bb.Emu = orig
    return out

plain, bf16 = run("plain"), run("bf16")
cos = lambda a, b: float(torch.nn.functional.cosine_similarity(a, b, dim=1).min())
print("embedding cosine, min over 32 rows, against the plain float64 forward: bf16 storage %.5f" % cos(bf16, plain))
for which, what in (("all", "every stored activation in e4m3"), ("conv", "convolution outputs in e4m3, BatchNorm outputs bf16"),
                    ("act", "BatchNorm / PReLU outputs in e4m3, convolution outputs bf16")):
    out = run(which)
    print("  %-62s %.5f (vs the bf16-storage forward %.5f)" % (what, cos(out, plain), cos(out, bf16)))
