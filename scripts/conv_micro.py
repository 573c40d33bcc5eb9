"""Micro-benchmark of one convolution shape (diagnostic; used under rocprofv3 --pmc)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops, _lib
variant = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cin = cout = int(sys.argv[3]) if len(sys.argv) > 3 else 256
hw = int(sys.argv[4]) if len(sys.argv) > 4 else 14
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
L = _lib.lib(); L.vlsfr_set_option(b"conv_glds", ctypes.c_int32(variant))
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); L.vlsfr_set_option(k.encode(), ctypes.c_int32(int(v)))
x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
ks = int(os.environ.get("KS", "3"))
w = (torch.randn(cout, ks, ks, cin, device="cuda") * 0.05).to(torch.bfloat16)
d = ops.ConvDesc(B, hw, hw, cin, cout, ks, ks, 1, ks // 2)
stats = ops.new_sums(cout, "cuda") if os.environ.get("NOSTATS") is None else None
mode = os.environ.get("MODE", "fwd")
if mode == "wgrad":
    dy = torch.randn(B, hw, hw, cout, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(cout, ks, ks, cin, device="cuda")
    run = lambda: ops.conv2d_wgrad(dy, x, d, dw=dw)
else:
    run = lambda: ops.conv2d_fwd(x, w, d, stats=stats)
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
fl = 2.0 * B * hw * hw * cout * ks * ks * cin
print("variant %d B=%d %dx%d@%d: %.1f us, %.1f TFLOP/s" % (variant, B, cin, cout, hw, dt * 1e6, fl / dt / 1e12))
