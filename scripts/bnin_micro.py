"""bn_apply + conv forward against the convolution that applies the BatchNorm (+ PReLU) in its operand path (vlsfr_conv2d_fwd_bnin),
per layer shape of ir100 at batch 256: microseconds per layer (back-to-back launches)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = _lib.lib()
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); L.vlsfr_set_option(k.encode(), ctypes.c_int32(int(v)))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for c, hw in ((256, 14), (128, 28)):
    d = ops.ConvDesc(B, hw, hw, c, c, 3, 3, 1, 1)
    M = B * hw * hw
    x = torch.randn(B, hw, hw, c, device="cuda").to(torch.bfloat16)
    w = (torch.randn(c, 3, 3, c, device="cuda") * 0.05).to(torch.bfloat16)
    gamma, beta, slope = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), torch.full((c,), 0.25, device="cuda")
    sums = ops.bn_stats(x, M, c)
    st = ops.new_sums(c, "cuda")
    for prelu in (False, True):
        sl = slope if prelu else None
        def separate():
            a, _, _ = ops.bn_apply(x, M, c, hw * hw, sums, gamma, beta, sl)
            ops.conv2d_fwd(a.view(B, hw, hw, c), w, d, stats=st)
        def fused(want_a):
            _, _, sc, sh = ops.bn_finalize(sums, M, c, gamma, beta)
            ops.conv2d_fwd_bnin(x, w, d, sc, sh, sl, want_a=want_a, stats=st)
        t_conv = timeit(lambda: ops.conv2d_fwd(x, w, d, stats=st))
        print("%d ch %dx%d prelu=%d: conv alone %.1f us | bn_apply + conv %.1f | fused, a written %.1f | fused, no a %.1f" %
              (c, hw, hw, prelu, t_conv, timeit(separate), timeit(lambda: fused(True)), timeit(lambda: fused(False))), flush=True)
