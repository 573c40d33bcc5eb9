"""Phase timing inside one workgroup of the ping-pong conv kernel (vlsfr_conv_trace stamps)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops, _lib
B, c, hw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L = _lib.lib(); L.vlsfr_set_option(b"conv_glds", ctypes.c_int32(9))
x = torch.randn(B, hw, hw, c, device="cuda").to(torch.bfloat16)
w = (torch.randn(c, 3, 3, c, device="cuda") * 0.05).to(torch.bfloat16)
d = ops.ConvDesc(B, hw, hw, c, c, 3, 3, 1, 1)
stats = ops.new_sums(c, "cuda")
for _ in range(3): ops.conv2d_fwd(x, w, d, stats=stats)
buf = torch.zeros(2, 64, 8, dtype=torch.int64, device="cuda")
L.vlsfr_conv_trace(ctypes.c_void_p(buf.data_ptr()))
ops.conv2d_fwd(x, w, d, stats=stats)
torch.cuda.synchronize()
L.vlsfr_conv_trace(ctypes.c_void_p(0))
t = buf.cpu().numpy()
nk = 9 * c // 64
names = {0: ["reads+issue", "wait lgkm", "barrier", "mfma", "wait vm", "barrier(next)"],
         1: ["reads+issue", "wait lgkm", "wait vm", "barrier", "mfma", "barrier(next)"]}
for g in (0, 1):
    tt = t[g, :min(nk, 64)]
    base = tt[0, 0]
    print("group", g, "first stamp", base, "total cycles", tt[-1, 5] - base, "per tile", (tt[-1, 5] - base) / len(tt))
    import numpy as np
    d_ = np.diff(np.concatenate([tt[:, :6], np.roll(tt[:, 0], -1)[:, None]], axis=1), axis=1)[1:-1]
    print("   mean cycles per section:", {n: float(v) for n, v in zip(names[g], d_.mean(0).round(0))})
    print("   tile 5 stamps:", (tt[5] - base)[:6], " tile 6:", (tt[6] - base)[:6])
