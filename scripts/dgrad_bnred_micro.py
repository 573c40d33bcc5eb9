"""Input-gradient convolution with / without the fused BatchNorm-backward reduction (vlsfr_conv2d_dgrad_bnred) against
the stand-alone reduction kernel it replaces, per distinct stride-1 / stride-2 3x3 shape of ir100 at batch_size 256."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = _lib.lib()
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); L.vlsfr_set_option(k.encode(), ctypes.c_int32(int(v)))
SHAPES = [(64, 64, 1, 56, 4), (128, 128, 1, 28, 24), (256, 256, 1, 14, 58), (512, 512, 1, 7, 4), (128, 128, 2, 56, 1), (256, 256, 2, 28, 1)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("%-22s | %9s %9s %9s | %9s %9s | %s" % ("Cin Cout s Hin", "dgrad us", "+red us", "+prelu us", "reduce us", "reduce-p", "saved per pair (us)"))
for cin, cout, stride, h, cnt in SHAPES:
    ho = ops.out_hw(h, 3, stride, 1)
    d = ops.ConvDesc(B, h, h, cin, cout, 3, 3, stride, 1)
    wf = torch.randn(cout, 3, 3, cin, device="cuda") * 0.05
    wT = wf.permute(3, 1, 2, 0).contiguous().to(torch.bfloat16)
    dy = torch.randn(B, ho, ho, cout, device="cuda").to(torch.bfloat16)
    x = torch.randn(B, h, h, cin, device="cuda").to(torch.bfloat16)
    M = B * h * h
    mean = torch.zeros(cin, device="cuda"); invstd = torch.ones(cin, device="cuda")
    gamma = torch.ones(cin, device="cuda"); beta = torch.zeros(cin, device="cuda"); slope = torch.full((cin,), 0.25, device="cuda")
    dx = ops.conv2d_dgrad(dy, wT, d)
    t0 = timeit(lambda: ops.conv2d_dgrad(dy, wT, d))
    t1 = timeit(lambda: ops.conv2d_dgrad_bnred(dy, wT, d, x, mean, invstd))
    t2 = timeit(lambda: ops.conv2d_dgrad_bnred(dy, wT, d, x, mean, invstd, gamma, beta, slope))
    r0 = timeit(lambda: ops.bn_backward_reduce(dx, x, M, cin, h * h, mean, invstd))
    r1 = timeit(lambda: ops.bn_backward_reduce(dx, x, M, cin, h * h, mean, invstd, gamma, beta, slope))
    print("%-22s | %9.1f %9.1f %9.1f | %9.1f %9.1f | %.1f / %.1f" % ("%d %d %d %d" % (cin, cout, stride, h), t0, t1, t2, r0, r1, t0 + r0 - t1, t0 + r1 - t2), flush=True)
