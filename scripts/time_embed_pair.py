"""Which ingredient of the distributed setup slows the two-backbone forward (probe on the main stream, gallery on a side
stream)?  Times FFC.embed_pair(x, y, update_gallery=False) (a) as built, (b) after the partitioned optimizer re-pointed the
parameters into its flat buffer, (c) after the process group (RCCL, one rank) and the ShardedFFC wrapper exist."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsfr_amd  # noqa
from vlsfr_amd.ffc import FFC

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
net = sys.argv[1] if len(sys.argv) > 1 else "ir100"
B = 256
torch.manual_seed(0)
m = FFC(net, 512, 65536, 32.0, "Arc", 0.5, 0.99, pool_device=dev).cuda()
x = torch.randn(B, 3, 112, 112, device=dev)
y = torch.randn(B, 3, 112, 112, device=dev)

def timeit(tag, n=8):
    for _ in range(2):
        p, g = m.embed_pair(x, y, False)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        p, g = m.embed_pair(x, y, False)
    b.record(); torch.cuda.synchronize()
    print("%-62s %.2f ms per pair" % (tag, a.elapsed_time(b) / n), flush=True)

timeit("(a) as built")
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
if "pg_first" in sys.argv:
    dist.init_process_group("nccl", device_id=dev)
    t = torch.ones(4, device=dev); dist.all_reduce(t)
    timeit("(a2) process group initialised, one all-reduce done")
from vlsfr_amd.optim.fused import PartitionedSGD
class Solo:
    world, rank = 1, 0
    def all_gather_into(self, out, shard): out.copy_(shard.reshape(-1)); return out
    def reduce_scatter_sum(self, out, inp): out.copy_(inp); return out
pn = m.probe_net
names = {id(p): n for n, p in pn.named_parameters()}
params = [p for p in m.parameters() if p.requires_grad]
opt = PartitionedSGD(params, 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True, comm=Solo(),
                     bucket_of=lambda p: pn.bucket_of(names[id(p)]), n_buckets=pn.N_BUCKETS)
opt.partition()
timeit("(b) parameters re-pointed into the partitioned flat buffer")
if not dist.is_initialized():
    dist.init_process_group("nccl", device_id=dev)
from vlsfr_amd.parallel import ShardedFFC
sm = ShardedFFC(m, dist)
timeit("(c) process group + ShardedFFC wrapper")
# one full distributed step (forward, backward, bucket reduce-scatters, partitioned update), then the pair again
lab = torch.from_numpy(np.random.default_rng(0).choice(60000, size=B, replace=False).astype(np.int64))
opt2 = sm.make_optimizer(0.1, 0.9, 1e-4, True) if "own_opt" in sys.argv else opt
for it in range(2):
    opt2.zero_grad()
    loss = sm(x, y, lab, lab)
    loss.backward()
    if hasattr(opt2, "reduce_bucket") and hasattr(opt2.comm, "dist"):
        sm.reduce_gradients(opt2)
    opt2.step()
torch.cuda.synchronize()
timeit("(c2) after two full distributed steps")
import gc
del loss
gc.collect()
torch.cuda.synchronize()
timeit("(c3) after dropping the last loss / graph")
pn.__dict__["signal_stages"] = False
timeit("(d) same, signal_stages off")
dist.destroy_process_group()
timeit("(e) process group destroyed")
