"""Do streaming kernels with few registers run BESIDE the one-wave-per-SIMD convolution (480 of a SIMD's 512 registers, all LDS)?
Stream A: N launches of the 256-channel 14x14 convolution (batch 256); stream B: M launches of the 10-register copy kernel
(vlsfr_copy_bytes, 51 MB read + 51 MB written each) — each alone, then both together."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops, _lib
L = _lib.lib()
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); L.vlsfr_set_option(k.encode(), ctypes.c_int32(int(v)))
B, C, H = 256, int(os.environ.get("C", 256)), int(os.environ.get("H", 14))
d = ops.ConvDesc(B, H, H, C, C, 3, 3, 1, 1)
x = torch.randn(B, H, H, C, device="cuda").to(torch.bfloat16)
w = (torch.randn(C, 3, 3, C, device="cuda") * 0.05).to(torch.bfloat16)
stats = ops.new_sums(C, "cuda")
src = torch.empty(B * H * H * C, dtype=torch.bfloat16, device="cuda").normal_()
dst = torch.empty_like(src)
nb = src.numel() * 2
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
N, M = 200, int(os.environ.get("M", 200))
def conv_loop():
    with torch.cuda.stream(sa):
        for _ in range(N): ops.conv2d_fwd(x, w, d, stats=stats)
MODE = os.environ.get("MODE", "copy")      # copy | bn (bn_apply, plain) | bnp (bn_apply with PReLU)
xs = src.view(B * H * H, C)
sums_b = ops.bn_stats(src, B * H * H, C)
gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda"); slope = torch.full((C,), 0.25, device="cuda")
def copy_loop(m):
    if MODE == "copy":
        for _ in range(m): L.vlsfr_copy_bytes(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), ctypes.c_size_t(nb), ctypes.c_void_p(sb.cuda_stream))
    else:
        with torch.cuda.stream(sb):
            for _ in range(m): ops.bn_apply(src, B * H * H, C, H * H, sums_b, gamma, beta, slope=slope if MODE == "bnp" else None)
def run(fa, fb):
    torch.cuda.synchronize(); ea0, ea1, eb0, eb1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t0 = time.perf_counter()
    ea0.record(sa); eb0.record(sb)
    if fa and fb:   # interleave the host-side issue so that both queues stay fed
        for i in range(10):
            with torch.cuda.stream(sa):
                for _ in range(N // 10): ops.conv2d_fwd(x, w, d, stats=stats)
            copy_loop(M // 10)
    elif fa: conv_loop()
    elif fb: copy_loop(M)
    ea1.record(sa); eb1.record(sb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, ea0.elapsed_time(ea1), eb0.elapsed_time(eb1)
for _ in range(2): run(True, True)
for rep in range(2):
    w_, a_, _ = run(True, False); print("conv alone : %d launches %.2f ms (%.1f us each)" % (N, a_, a_ * 1e3 / N))
    w_, _, b_ = run(False, True); print("copy alone : %d launches %.2f ms (%.1f us each, %.2f TB/s)" % (M, b_, b_ * 1e3 / M, 2 * nb * M / b_ / 1e9))
    w_, a_, b_ = run(True, True); print("both       : wall %.2f ms | conv stream %.2f ms (%.1f us each) | copy stream %.2f ms (%.1f us each)" % (w_, a_, a_ * 1e3 / N, b_, b_ * 1e3 / M))
