"""Micro-benchmark of the BatchNorm backward kernels on one tensor shape (diagnostic)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops, _lib
kb = int(sys.argv[1]); M = int(sys.argv[2]); C = int(sys.argv[3])
L = _lib.lib(); L.vlsfr_set_option(b"bn_block_kb", ctypes.c_int32(kb))
x = torch.randn(M, C, device="cuda").to(torch.bfloat16); dy = torch.randn(M, C, device="cuda").to(torch.bfloat16)
g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"); sl = torch.full((C,), 0.25, device="cuda")
sums = ops.bn_stats(x, M, C)
y, mean, invstd = ops.bn_apply(x, M, C, 196, sums, g, b, sl)
dg, db, ds = (torch.zeros(C, device="cuda") for _ in range(3))
def run():
    ops.bn_backward(dy, x, M, C, 196, mean, invstd, g, b, sl, None, dg, db, ds)
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("bn_backward kb=%d M=%d C=%d: %.1f us (5 tensor passes = %.2f TB/s)" % (kb, M, C, dt * 1e6, 5 * M * C * 2 / dt / 1e12))
