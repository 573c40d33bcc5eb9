"""Diagnostic (not a test): per-parameter gradient error of the GPU backbone vs the float64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ffc_ref, backbones_ref as bb
from tests.golden import common
from vlsfr_amd.ffc import FFC

net, B = sys.argv[1], int(sys.argv[2])
Q, D = 96, 64
o = ffc_ref.FFCRef(net, D, Q, 32.0, "AM", 0.4, 0.99, dtype=torch.float64, layers=(1,1,1,1) if net=="irtiny" else None)
sd = common.fill_state({k: v.detach() for k, v in o.probe.items()}, 77)
o.probe = {k: (v.double().requires_grad_(bb.trainable(k)) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
o.gallery = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
m = FFC(net, D, Q, 32.0, "AM", 0.4, 0.99, precise_head=True)
m.probe_net.load_state_dict(sd); m.gallery_net.load_state_dict(sd)
m = m.cuda(); m.queue.copy_(o.queue.float())
rng = np.random.default_rng(9)
xu8, yu8 = common.synth_images_u8(rng, B), common.synth_images_u8(rng, B)
ids = rng.choice(50, size=B // 2, replace=False)
xl = torch.from_numpy(np.concatenate([ids, rng.integers(0, 50, B // 2)]).astype(np.int64))
yl = torch.from_numpy(np.concatenate([ids, rng.integers(0, 50, B // 2)]).astype(np.int64))
x, y = common.images_from_u8(xu8), common.images_from_u8(yu8)
lo = o.forward(x.double(), y.double(), xl, yl); lo.backward()
lg = m(x.cuda(), y.cuda(), xl, yl); lg.backward()
print("loss", float(lo), float(lg))
for n, p in m.probe_net.named_parameters():
    if not p.requires_grad: continue
    g, w = p.grad.detach().cpu().double().numpy(), o.probe[n].grad.numpy()
    nw = np.linalg.norm(w)
    print("%-34s |g|=%10.4g rel_l2=%.4f" % (n, nw, np.linalg.norm(g - w) / (nw + 1e-30)))
