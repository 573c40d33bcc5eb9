"""Two dependent chains (conv -> bn_apply -> conv -> bn_apply ...) as in the two backbone passes of a step: one stream against two.
How much of the BatchNorm time hides under the other chain's convolutions?"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd import ops, _lib
L = _lib.lib()
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); L.vlsfr_set_option(k.encode(), ctypes.c_int32(int(v)))
B, C, H = 256, int(os.environ.get("C", 256)), int(os.environ.get("H", 14))
NL = int(os.environ.get("NL", 60))
d = ops.ConvDesc(B, H, H, C, C, 3, 3, 1, 1)
M = B * H * H
class Chain:
    def __init__(self):
        self.x = torch.randn(B, H, H, C, device="cuda").to(torch.bfloat16)
        self.w = (torch.randn(C, 3, 3, C, device="cuda") * 0.02).to(torch.bfloat16)
        self.gamma = torch.ones(C, device="cuda"); self.beta = torch.zeros(C, device="cuda")
        self.stats = [ops.new_sums(C, "cuda") for _ in range(NL)]
    def run(self):
        a = self.x
        for i in range(NL):
            self.stats[i].zero_()
        for i in range(NL):
            y = ops.conv2d_fwd(a, self.w, d, stats=self.stats[i])
            a, _, _ = ops.bn_apply(y, M, C, H * H, self.stats[i], self.gamma, self.beta)
        return a
c1, c2 = Chain(), Chain()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both(two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s1): c1.run()
    with torch.cuda.stream(s2 if two else s1): c2.run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
def one():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s1): c1.run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for _ in range(2): both(True); both(False)
for rep in range(3):
    print("one chain %.2f ms | two chains, one stream %.2f ms | two chains, two streams %.2f ms   (%d layers of conv + bn_apply, C=%d H=%d)" % (one(), both(False), both(True), NL, C, H))
