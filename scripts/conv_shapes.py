"""Per-shape time of the distinct convolutions of ir100 at batch_size 256 (forward, input gradient, weight gradient), weighted
by how often each runs in a training step (4 forward passes, 2 backward passes): where the conv milliseconds are."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = _lib.lib()
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); L.vlsfr_set_option(k.encode(), ctypes.c_int32(int(v)))
# (Cin, Cout, k, stride, Hin, count in ir100)   SURVEY 8(a-conv)
SHAPES = [(64, 64, 3, 1, 112, 1), (64, 64, 3, 2, 112, 1), (64, 64, 1, 2, 112, 1), (64, 64, 3, 1, 56, 4), (64, 128, 3, 1, 56, 1),
          (128, 128, 3, 2, 56, 1), (64, 128, 1, 2, 56, 1), (128, 128, 3, 1, 28, 24), (128, 256, 3, 1, 28, 1), (256, 256, 3, 2, 28, 1),
          (128, 256, 1, 2, 28, 1), (256, 256, 3, 1, 14, 58), (256, 512, 3, 1, 14, 1), (512, 512, 3, 2, 14, 1), (256, 512, 1, 2, 14, 1),
          (512, 512, 3, 1, 7, 4)]
if os.environ.get("ONLY"):
    keep = set(os.environ["ONLY"].split())
    SHAPES = [sh for sh in SHAPES if "%d_%d_%d_%d_%d" % sh[:5] in keep]
def timeit(fn, n=int(os.environ.get('ITERS', 10))):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print("%-28s %5s | %9s %7s | %9s %7s | %9s %7s | ms/step" % ("Cin Cout k s Hin", "count", "fwd us", "TF/s", "dgrad us", "TF/s", "wgrad us", "TF/s"))
for cin, cout, k, stride, h, cnt in SHAPES:
    pad = k // 2
    ho = ops.out_hw(h, k, stride, pad)
    d = ops.ConvDesc(B, h, h, cin, cout, k, k, stride, pad)
    x = torch.randn(B, h, h, cin, device="cuda").to(torch.bfloat16)
    wf = torch.randn(cout, k, k, cin, device="cuda") * 0.05
    w = wf.to(torch.bfloat16)
    wT = wf.permute(3, 1, 2, 0).contiguous().to(torch.bfloat16)          # [Cin][R][S][Cout]
    dy = torch.randn(B, ho, ho, cout, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(cout, k, k, cin, device="cuda")
    stats = ops.new_sums(cout, "cuda")
    fl = 2.0 * B * ho * ho * cout * k * k * cin
    tf = timeit(lambda: ops.conv2d_fwd(x, w, d, stats=None if os.environ.get("NOSTATS") else stats))
    if os.environ.get('BOTH'): print('   forward without statistics %.1f us' % (timeit(lambda: ops.conv2d_fwd(x, w, d, stats=None)) * 1e6))
    td = timeit(lambda: ops.conv2d_dgrad(dy, wT, d))
    tw = timeit(lambda: ops.conv2d_wgrad_ws(dy, x, d, dw=dw)) if os.environ.get("WS") else timeit(lambda: ops.conv2d_wgrad(dy, x, d, dw=dw))
    ms = cnt * (4 * tf + 2 * td + 2 * tw) * 1e3
    tot["fwd"] += cnt * 4 * tf * 1e3; tot["dgrad"] += cnt * 2 * td * 1e3; tot["wgrad"] += cnt * 2 * tw * 1e3
    print("%-28s %5d | %9.1f %7.0f | %9.1f %7.0f | %9.1f %7.0f | %6.2f" % ("%d %d %d %d %d" % (cin, cout, k, stride, h), cnt, tf * 1e6, fl / tf / 1e12,
                                                                      td * 1e6, fl / td / 1e12, tw * 1e6, fl / tw / 1e12, ms), flush=True)
print("per step: forward %.1f ms, input gradients %.1f ms, weight gradients %.1f ms" % (tot["fwd"], tot["dgrad"], tot["wgrad"]))
