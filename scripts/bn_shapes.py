"""Per-shape time of the BatchNorm kernels of ir100 at batch_size B (forward apply: plain / PReLU / residual + statistics of the
output; backward: reduce + apply), against a copy of the same bytes (vlsfr_copy_bytes)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = _lib.lib()
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); L.vlsfr_set_option(k.encode(), ctypes.c_int32(int(v)))
N = int(os.environ.get("ITERS", 40))
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / N * 1e6
print("%-14s %8s | %9s %9s %9s | %9s %9s | %9s" % ("C x H", "MB", "apply", "prelu", "res+stat", "bwd", "bwd prelu", "copy"))
for C, H in [(64, 56), (128, 28), (256, 14), (512, 7)]:
    M = B * H * H
    x = torch.randn(M * C, device="cuda").to(torch.bfloat16)
    r = torch.randn(M * C, device="cuda").to(torch.bfloat16)
    dy = torch.randn(M * C, device="cuda").to(torch.bfloat16)
    dst = torch.empty_like(x)
    sums = ops.bn_stats(x, M, C)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda"); sl = torch.full((C,), 0.25, device="cuda")
    osum = ops.new_sums(C, "cuda")
    _, mean, invstd = ops.bn_apply(x, M, C, H * H, sums, g, b)
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda"); dsl = torch.zeros(C, device="cuda")
    t0 = timeit(lambda: ops.bn_apply(x, M, C, H * H, sums, g, b))
    t1 = timeit(lambda: ops.bn_apply(x, M, C, H * H, sums, g, b, slope=sl))
    t2 = timeit(lambda: ops.bn_apply(x, M, C, H * H, sums, g, b, residual=r, out_sums=osum))
    t3 = timeit(lambda: ops.bn_backward(dy, x, M, C, H * H, mean, invstd, g, b, dgamma=dg, dbeta=db))
    t4 = timeit(lambda: ops.bn_backward(dy, x, M, C, H * H, mean, invstd, g, b, slope=sl, dgamma=dg, dbeta=db, dslope=dsl))
    nb = x.numel() * 2
    tc = timeit(lambda: L.vlsfr_copy_bytes(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(dst.data_ptr()), ctypes.c_size_t(nb), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    print("%-14s %8.1f | %9.1f %9.1f %9.1f | %9.1f %9.1f | %9.1f" % ("%d x %d" % (C, H), nb / 1e6, t0, t1, t2, t3, t4, tc), flush=True)
