"""Depthwise weight-gradient timing for several block counts (diagnostic; run under rocprofv3 --kernel-trace)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops, _lib
L = _lib.lib()
for nb in (64, 128, 256, 512):
    L.vlsfr_set_option(b"dw_wgrad_blocks", ctypes.c_int32(nb))
    for C, hw in ((64, 56), (128, 28), (256, 14)):
        B = 256
        x = torch.randn(B, hw, hw, C, device="cuda").to(torch.bfloat16); dy = torch.randn(B, hw, hw, C, device="cuda").to(torch.bfloat16)
        d = ops.ConvDesc(B, hw, hw, C, C, 3, 3, 1, 1)
        dw = torch.zeros(C, 3, 3, device="cuda")
        for _ in range(3): ops.dwconv_wgrad(dy, x, d, dw)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.dwconv_wgrad(dy, x, d, dw)
        e1.record(); torch.cuda.synchronize()
        print("blocks %4d C=%3d hw=%2d: %.1f us" % (nb, C, hw, e0.elapsed_time(e1) * 100))
