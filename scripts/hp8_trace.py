"""Clock stamps inside one workgroup of conv_igemm_hp8_kernel (vlsfr_conv_trace; DIAG = 1 instantiation): per phase of a
k-tile, the length of the section that reads LDS / issues DMA (incl. the return of its reads: the stamp waits on lgkmcnt), the
wait at the barrier behind it, the MFMA section and the wait at its closing barrier; waves 0 (first group) and 4 (second).
    python scripts/hp8_trace.py <batch> <channels> <hw>"""
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
for _ in range(200): ops.conv2d_fwd(x, w, d, stats=stats)      # warm: the clock the chip holds under this load
buf = torch.zeros(2, 64, 16, dtype=torch.int64, device="cuda")
L.vlsfr_conv_trace(ctypes.c_void_p(buf.data_ptr()))
for _ in range(3): ops.conv2d_fwd(x, w, d, stats=stats)
torch.cuda.synchronize()
L.vlsfr_conv_trace(ctypes.c_void_p(0))
t = buf.cpu().numpy()
nk = min(9 * c // 64, 64)
for g in (0, 1):
    tt = t[g, :nk].reshape(-1)                                    # stamps in program order: per phase [R start, R done, barrier passed, MFMAs issued]
    if tt[0] == 0:
        print("group", g, "no stamps (was the kernel taken? tiles >= 8?)"); continue
    dd = np.diff(tt)[: 16 * (nk - 1)].reshape(nk - 1, 16)[2:]    # drop the first two k-tiles (prologue effects)
    names = ["R", "barrier A", "M", "barrier B"]
    print("group %d: cycles per k-tile %.0f (stamps included)" % (g, dd.sum(1).mean()))
    for ph in range(4):
        print("   phase %d: " % (ph + 1) + "  ".join("%s %.0f" % (names[k], dd[:, ph * 4 + k].mean()) for k in range(4)))
