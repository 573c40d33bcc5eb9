import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.golden import common
from vlsfr_amd import _lib
from vlsfr_amd.ffc import FFC
L = _lib.lib()
for kv in os.environ.get("OPTS", "").split():
    k, v = kv.split("="); L.vlsfr_set_option(k.encode(), ctypes.c_int32(int(v)))
torch.manual_seed(11)
m = FFC("irtiny", 512, 4096, 32.0, "Arc", 0.5, 0.99).cuda()
rng = np.random.default_rng(5)
B = 32
x = common.images_from_u8(common.synth_images_u8(rng, B)).cuda()
y = common.images_from_u8(common.synth_images_u8(rng, B)).cuda()
ids = rng.choice(6000, size=B // 2, replace=False)
xl = torch.from_numpy(np.concatenate([ids, rng.integers(0, 6000, B - B // 2)]).astype(np.int64))
yl = torch.from_numpy(np.concatenate([ids, rng.integers(0, 6000, B - B // 2)]).astype(np.int64))
for it in range(2):
    m.zero_grad()
    loss = m(x, y, xl, yl)
    loss.backward()
    torch.cuda.synchronize()
    bad = [k for k, p in m.probe_net.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    print("iter", it, "loss", float(loss), "non-finite grads:", len(bad), bad[:12], flush=True)
