"""Times the stages of one identity-sharded head pass (world 1 over RCCL) against the single-pool head: where does the
distributed step spend its extra milliseconds?  usage: python scripts/time_sharded_head.py [Q] [B]"""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsfr_amd  # noqa
from vlsfr_amd.ffc import build_pool
from vlsfr_amd.head import DcpHead, ShardedDcpHead
from vlsfr_amd.parallel import Comm

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 10 << 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
comm = Comm(dist)
queue = build_pool(Q, 512, dev, None, seed=0)
rng = np.random.default_rng(0)
ar = np.arange(Q)
sh = ShardedDcpHead(queue, 0, 1, Q, 32.0, 0.5, "Arc")
sh.lru.restore_arrays(ar.astype(np.int64), ar.astype(np.int32))
single = DcpHead(queue, 32.0, 0.5, "Arc")
single.lru.restore_arrays(ar.astype(np.int64), ar.astype(np.int32))
p = torch.nn.functional.normalize(torch.randn(B, 512, device=dev), dim=1)
g = torch.nn.functional.normalize(torch.randn(B, 512, device=dev), dim=1)
lab = rng.choice(Q, size=B, replace=False).astype(np.int64)

def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e

for rep in range(4):
    t = [ev()]
    with torch.no_grad():
        pg = comm.all_gather(torch.cat([p, g], dim=1)).reshape(-1, 1024); t.append(ev())
        st = sh.begin(pg[:, :512].contiguous(), pg[:, 512:].contiguous(), lab, lab, True); t.append(ev())
        st = sh.sweep(st); t.append(ev())
        st = sh.combine(st, comm, own_rows=(0, B)); t.append(ev())
        loss, dP = sh.finish(st); t.append(ev())
    pp = p.clone().requires_grad_(True)
    l1 = single.run_pass(pp, g, lab, lab, True); t.append(ev())
    torch.cuda.synchronize()
    names = ["all_gather(p|g)", "begin (bookkeeping + H2D)", "sweep (shard partial)", "combine (collectives + rescale)", "finish", "single-pool run_pass"]
    if rep:
        print("rep %d: " % rep + "; ".join("%s %.2f ms" % (n, t[i].elapsed_time(t[i + 1])) for i, n in enumerate(names)),
              "| loss sharded %.4f single %.4f" % (float(loss), float(l1)))
dist.destroy_process_group()
