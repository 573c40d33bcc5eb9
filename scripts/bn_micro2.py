"""Micro-benchmark of bn_apply / bn_backward over the ir50 layer shapes at batch B (run under rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
import ctypes
from vlsfr_amd import _lib
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); _lib.lib().vlsfr_set_option(k.encode(), ctypes.c_int32(int(v)))
shapes = ((112, 64), (56, 64), (56, 128), (28, 128), (28, 256), (14, 256), (14, 512), (7, 512))
if os.environ.get("SHAPES"): shapes = [tuple(int(t) for t in sh.split("x")) for sh in os.environ["SHAPES"].split()]
for hw, C in shapes:
    M = B * hw * hw
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16); dy = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"); sl = torch.full((C,), 0.25, device="cuda")
    dg, db, ds = (torch.zeros(C, device="cuda") for _ in range(3))
    for _ in range(5):
        sums = ops.bn_stats(x, M, C)
        y, mean, invstd = ops.bn_apply(x, M, C, hw * hw, sums, g, b, sl)
        ops.bn_backward(dy, x, M, C, hw * hw, mean, invstd, g, b, sl, None, dg, db, ds)
    torch.cuda.synchronize()
    print("shape", hw, C, "MB", M * C * 2 / 1e6)
