"""Depthwise 3x3 kernels of MobileFaceNet at batch_size 256: time and streamed GB/s (input + output bytes) per shape,
forward / input gradient / weight gradient, weighted by layer count and passes per step (4 forward, 2 backward)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(64, 56, 1, 1), (128, 56, 2, 1), (128, 28, 1, 4), (256, 28, 2, 1), (256, 14, 1, 6), (512, 14, 2, 1), (256, 7, 1, 2)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
tot = 0.0
for C, h, s, cnt in SHAPES:
    ho = ops.out_hw(h, 3, s, 1)
    d = ops.ConvDesc(B, h, h, C, C, 3, 3, s, 1)
    x = torch.randn(B, h, h, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(C, 3, 3, device="cuda") * 0.2)
    dy = torch.randn(B, ho, ho, C, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(C, 3, 3, device="cuda")
    st = ops.new_sums(C, "cuda")
    bi, bo = B * h * h * C * 2, B * ho * ho * C * 2
    tf = timeit(lambda: ops.dwconv_fwd(x, w, d, stats=st))
    td = timeit(lambda: ops.dwconv_dgrad(dy, w, d))
    tw = timeit(lambda: ops.dwconv_wgrad_ws(dy, x, d, dw))
    ms = cnt * (4 * tf + 2 * td + 2 * tw) * 1e3
    tot += ms
    print("C=%3d %2dx%2d s%d x%d | fwd %6.1f us %5.2f TB/s | dgrad %6.1f us %5.2f TB/s | wgrad %6.1f us %5.2f TB/s | %.2f ms/step" %
          (C, h, h, s, cnt, tf * 1e6, (bi + bo) / tf / 1e12, td * 1e6, (bi + bo) / td / 1e12, tw * 1e6, (bi + bo) / tw / 1e12, ms), flush=True)
print("depthwise kernels per step: %.1f ms" % tot)
# the 7x7 valid "linear7" layer (mobilefacenet_def.py:88): [B, 7, 7, 512] -> [B, 1, 1, 512]
d = ops.ConvDesc(B, 7, 7, 512, 512, 7, 7, 1, 0)
x = torch.randn(B, 7, 7, 512, device="cuda").to(torch.bfloat16)
w = torch.randn(512, 49, device="cuda") * 0.1
dy = torch.randn(B, 1, 1, 512, device="cuda").to(torch.bfloat16)
dw = torch.zeros(512, 49, device="cuda")
st = ops.new_sums(512, "cuda")
tf = timeit(lambda: ops.dwconv_fwd(x, w, d, stats=st)); td = timeit(lambda: ops.dwconv_dgrad(dy, w, d)); tw = timeit(lambda: ops.dwconv_wgrad(dy, x, d, dw))
print("linear7 512 7x7 valid | fwd %6.1f us | dgrad %6.1f us | wgrad %6.1f us | %.2f ms/step" % (tf * 1e6, td * 1e6, tw * 1e6, (4 * tf + 2 * td + 2 * tw) * 1e3))
