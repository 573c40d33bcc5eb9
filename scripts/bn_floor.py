"""BatchNorm kernels against a plain device copy of the same bytes, per activation shape of ir100 at batch_size B
(where are the normalisation kernels relative to what the part sustains for tensors of this size?)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); _lib.lib().vlsfr_set_option(k.encode(), ctypes.c_int32(int(v)))
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("%-12s %7s | %8s %8s | %8s %8s %8s | %8s %8s" % ("hw x C", "MB", "copy us", "TB/s", "apply0", "apply1", "apply6", "bwd0 us", "bwd12 us"))
for hw, C in ((56, 64), (28, 128), (14, 256), (7, 512)):
    M = B * hw * hw
    nb = M * C * 2
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16); dy = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    r = torch.randn(M, C, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"); sl = torch.full((C,), 0.25, device="cuda")
    dg, db, ds = (torch.zeros(C, device="cuda") for _ in range(3))
    sums = ops.bn_stats(x, M, C)
    osums = ops.new_sums(C, "cuda")
    _, mean, invstd = ops.bn_apply(x, M, C, hw * hw, sums, g, b)
    tc = timeit(lambda: y.copy_(x))
    t0 = timeit(lambda: ops.bn_apply(x, M, C, hw * hw, sums, g, b))
    t1 = timeit(lambda: ops.bn_apply(x, M, C, hw * hw, sums, g, b, sl))
    t6 = timeit(lambda: ops.bn_apply(x, M, C, hw * hw, sums, g, b, None, r, out_sums=osums))
    b0 = timeit(lambda: ops.bn_backward(dy, x, M, C, hw * hw, mean, invstd, g, b, None, None, dg, db, None))
    b12 = timeit(lambda: ops.bn_backward(dy, x, M, C, hw * hw, mean, invstd, g, b, None, r, dg, db, None))
    print("%-12s %7.1f | %8.1f %8.2f | %8.1f %8.1f %8.1f | %8.1f %8.1f" % ("%dx%d x %d" % (hw, hw, C), nb / 1e6, tc, 2 * nb / tc / 1e6, t0, t1, t6, b0, b12), flush=True)
