"""Which Python call sites issue the small device-to-device copies of a step (diagnostic, torch.profiler with stacks)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vlsfr_amd.ffc import FFC
from vlsfr_amd.optim import get_optim_scheduler
torch.manual_seed(0)
Q = 1 << 16
m = FFC("ir50", 512, Q, 32.0, "Arc", 0.5, 0.99).cuda()
m.lru.restore(list(zip(range(Q), range(Q))))
m.concurrent_streams = False; m.probe_net.concurrent_backward = False
cfg = dict(optim="SGD", scheduler="multistep", LR=0.1, momentum=0.9, decay=1e-4, nesterov=True, warmup=0, epochs=1, milestones=[8], gammas=[0.1])
opt, sched = get_optim_scheduler([p for p in m.parameters() if p.requires_grad], cfg); sched.update(0, 0.0)
rng = np.random.default_rng(0)
x, y, xl, yl = bench.synth_batch(rng, 32, Q, torch.device("cuda"))
def step():
    opt.zero_grad(); loss = m(x, y, xl, yl); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::to", "aten::_to_copy"):
        st = [s for s in (ev.stack or []) if "very-large" in s or "bench" in s or "vlsfr" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for k, v in cnt.most_common(12): print(v, k)
