"""Shader-clock stamps of wave 0 of one workgroup of conv_igemm_hw4_kernel (DIAG = 1 instantiation, vlsfr_conv_trace): cycles of the
two steps of every k-tile, and the clock the chip holds (shader cycles over the loop / 100 MHz real-time ticks).
    python scripts/hw4_trace.py <batch> <channels> <hw>"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vlsfr_amd import ops, _lib
B, c, hw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L = _lib.lib()
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); L.vlsfr_set_option(k.encode(), ctypes.c_int32(int(v)))
x = torch.randn(B, hw, hw, c, device="cuda").to(torch.bfloat16)
w = (torch.randn(c, 3, 3, c, device="cuda") * 0.05).to(torch.bfloat16)
d = ops.ConvDesc(B, hw, hw, c, c, 3, 3, 1, 1)
stats = ops.new_sums(c, "cuda")
for _ in range(300): ops.conv2d_fwd(x, w, d, stats=stats)      # warm: the clock the chip holds under this load
buf = torch.zeros(2, 64, 16, dtype=torch.int64, device="cuda")
L.vlsfr_conv_trace(ctypes.c_void_p(buf.data_ptr()))
for _ in range(20): ops.conv2d_fwd(x, w, d, stats=stats)
torch.cuda.synchronize()
L.vlsfr_conv_trace(ctypes.c_void_p(0))
t = buf.cpu().numpy().reshape(-1)
nk = min(9 * c // 64, 64)
st = t[: 2 * nk].reshape(nk, 2)
if st[0, 0] == 0:
    sys.exit("no stamps")
a = st[1:, 0] - st[:-1, 0]                       # k-tile to k-tile
s0 = st[:, 1] - st[:, 0]                         # step 2u (+ the wait and the barrier at the head of step 2u + 1)
s1 = st[1:, 0] - st[:-1, 1]                      # step 2u + 1
real = (t[129] - t[128]) / 100e6                 # seconds
cyc = st[-1, 1] - st[0, 0]
print("k-tiles %d; cycles per k-tile: mean %.0f min %d max %d | step 0 (+ barrier) %.0f, step 1 %.0f  (ideal 896 each)" %
      (nk, a.mean(), a.min(), a.max(), s0[1:-1].mean(), s1[1:-1].mean()))
print("loop %.2f us by the real-time clock; shader clock ~ %.2f GHz" % (real * 1e6, cyc / real / 1e9 if real > 0 else 0))
print("in-kernel (100 MHz clock): entry -> loop %.2f us, loop %.2f us, loop end -> stores acknowledged %.2f us" %
      ((t[128] - t[130]) / 100.0, (t[129] - t[128]) / 100.0, (t[131] - t[129]) / 100.0))
print("per k-tile:", " ".join(str(int(v)) for v in a[:36]))
