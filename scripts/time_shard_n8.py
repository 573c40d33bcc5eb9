"""Rank 0's share of the head at the node size the metric is quoted on (8 ranks x batch_size 256, 10 485 760 identities):
its 1 310 720 slots against all 2048 gathered rows — begin + sweep (what needs no peer), for bf16 and fp8, Arc and SV.
Checks the launch geometry at these shapes (16 row blocks) and prints the times."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsfr_amd  # noqa
from vlsfr_amd.ffc import build_pool
from vlsfr_amd.head import ShardedDcpHead

world, B = 8, 256
Q = 10 << 20
Qs = Q // world
dev = torch.device("cuda", 0)
queue = build_pool(Q, 512, dev, (0, world), seed=0)
rng = np.random.default_rng(0)
rows = world * B
p = torch.nn.functional.normalize(torch.randn(rows, 512, device=dev), dim=1)
g = torch.nn.functional.normalize(torch.randn(rows, 512, device=dev), dim=1)
ar = np.arange(Q)
for loss_type, margin in (("Arc", 0.5), ("SV", 0.35)):
    for dtype in ("bf16", "fp8"):
        h = ShardedDcpHead(queue, 0, world, Q, 32.0, margin, loss_type)
        h.head_dtype = dtype
        h.lru.restore_arrays(ar.astype(np.int64), ar.astype(np.int32))
        lab = rng.choice(Q, size=rows, replace=False).astype(np.int64)
        lab[::7] = Q + 5 + np.arange(len(lab[::7]))            # some identities outside the pool: outlier rows (top-k path)
        for rep in range(3):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record()
            st = h.begin(p, g, lab, lab, True)
            st = h.sweep(st)
            b.record(); torch.cuda.synchronize()
        ok = all(bool(torch.isfinite(st[k]).all()) for k in ("M", "packed"))
        print("%-3s %-4s: begin + sweep of 1 310 720 slots x 2048 rows %.2f ms; finite %s; own-label rows %d" %
              (loss_type, dtype, a.elapsed_time(b), ok, int((st["label"] >= 0).sum())), flush=True)
