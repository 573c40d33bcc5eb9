"""Full-step time of the single-process path and of the distributed path's ingredients added one at a time (one rank):
which one costs the ~10 ms per step that `bench.py --force-dist` shows?"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsfr_amd  # noqa
from vlsfr_amd.ffc import FFC
from vlsfr_amd.optim.fused import FusedSGD, PartitionedSGD

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, Q = 256, 1 << 20
x = torch.randn(B, 3, 112, 112, device=dev); y = torch.randn(B, 3, 112, 112, device=dev)
rng = np.random.default_rng(0)
lab = torch.from_numpy(rng.choice(Q, size=B, replace=False).astype(np.int64))

def build():
    torch.manual_seed(0)
    m = FFC("ir100", 512, Q, 32.0, "Arc", 0.5, 0.99, pool_device=dev).cuda()
    ar = np.arange(Q)
    m.lru.restore_arrays(ar.astype(np.int64), ar.astype(np.int32))
    return m

class Solo:
    world, rank = 1, 0
    def all_gather_into(self, out, shard): out.copy_(shard.reshape(-1)); return out
    def reduce_scatter_sum(self, out, inp): out.copy_(inp); return out

def run(tag, step, n=6):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print("%-70s %.1f ms per step" % (tag, (time.perf_counter() - t0) / n * 1e3), flush=True)

which = sys.argv[1:] or ["v0", "v1", "v2", "v3"]
if "v0b" in which:    # the single-process path at the metric's pool size
    QB = int(os.environ.get("VB_Q", 10 << 20))
    torch.manual_seed(1234)
    m = FFC("ir100", 512, QB, 32.0, "Arc", 0.5, 0.99, pool_device=dev).cuda()
    ar = np.arange(QB)
    m.lru.restore_arrays(ar.astype(np.int64), ar.astype(np.int32))
    opt = FusedSGD([p for p in m.parameters() if p.requires_grad], 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    def s0b():
        opt.zero_grad(); loss = m(x, y, lab, lab); loss.backward(); opt.step()
    run("v0b single-process path, 10 M slots", s0b)
    del m, opt
if "v0" in which:
    m = build()
    opt = FusedSGD([p for p in m.parameters() if p.requires_grad], 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    def s0():
        opt.zero_grad(); loss = m(x, y, lab, lab); loss.backward(); opt.step()
    run("v0 single-process path (FFC + FusedSGD)", s0)
    del m, opt
if "v5" in which:
    m = build()
    opt = FusedSGD([p for p in m.parameters() if p.requires_grad], 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    head = m._ensure_head()
    ln = lab.numpy()
    def s5():
        opt.zero_grad()
        p1, g1 = m.embed_pair(x, y, True)
        l2 = head.run_pass(p1, g1, ln, ln, True)
        p2, g2 = m.embed_pair(y, x, False)
        l1 = head.run_pass(p2, g2, ln, ln, False)
        (l1 + l2).backward(); opt.step()
    run("v5 FFC pieces in the wrapper's order: heads on the main stream", s5)
    del m, opt, head
if "v1" in which or "v2" in which:
    m = build()
    pn = m.probe_net
    names = {id(p): n for n, p in pn.named_parameters()}
    opt = PartitionedSGD([p for p in m.parameters() if p.requires_grad], 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True,
                         comm=Solo(), bucket_of=lambda p: pn.bucket_of(names[id(p)]), n_buckets=pn.N_BUCKETS)
    def s1():
        opt.zero_grad(); loss = m(x, y, lab, lab); loss.backward()
        for b in range(opt.n_buckets):
            opt.reduce_bucket(b)
        opt.step()
    if "v1" in which:
        run("v1 FFC + PartitionedSGD (local stand-in for the collectives)", s1)
    if "v2" in which:
        pn.__dict__["signal_stages"] = True
        run("v2 = v1 + staged backward (per-bucket events recorded)", s1)
        pn.__dict__["signal_stages"] = False
    del m, opt
if any(k in which for k in ("v3", "v3a", "vb", "w")):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29537")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=dev)
    from vlsfr_amd.parallel import ShardedFFC, DataParallelFFC
    if "v3a" in which:
        m = build()
        opt = FusedSGD([p for p in m.parameters() if p.requires_grad], 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
        def s3a():
            opt.zero_grad(); loss = m(x, y, lab, lab); loss.backward(); opt.step()
        run("v3a process group (RCCL, device_id) initialised; single-process step", s3a)
        t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
        run("v3a' same after one all-reduce", s3a)
        g = dist.new_group(backend="gloo")
        run("v3a'' same after creating the gloo side group", s3a)
        del m, opt
    m = sm = opt = None
    def s3():
        opt.zero_grad(); loss = sm(x, y, lab, lab); loss.backward(); sm.reduce_gradients(opt); opt.step()
    if "v3" in which:
        m = build()
        sm = ShardedFFC(m, dist)
        opt = sm.make_optimizer(0.1, 0.9, 1e-4, True)
        run("v3 ShardedFFC + PartitionedSGD over RCCL (one rank)", s3)
    if "vb" in which:     # bench.py --force-dist as closely as possible: shard-local pool of the metric's size, LR schedule
        QB = int(os.environ.get("VB_Q", 10 << 20))
        torch.manual_seed(1234)
        m = FFC("ir100", 512, QB, 32.0, "Arc", 0.5, 0.99, pool_device=dev, pool_shard=(0, 1)).cuda()
        ar = np.arange(QB)
        m.lru.restore_arrays(ar.astype(np.int64), ar.astype(np.int32))
        sm = ShardedFFC(m, dist)
        opt = sm.make_optimizer(0.1, 0.9, 1e-4, True)
        from vlsfr_amd.optim.optimizer import WarmupSchedule
        sched = WarmupSchedule(opt, "multistep", 0, 1, milestones=[8, 14, 17], gammas=[0.1, 0.1, 0.1])
        sched.update(0, 0.0)
        run("vb bench.py --force-dist rebuilt here (10 M slots, shard-local pool, schedule)", s3)
        if "data" in which:     # bench.py's synthetic batches (uniform uint8 pixels, id + instance label halves), rotating
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import bench as _b
            rngb = np.random.default_rng(1234)
            batches = [_b.synth_batch(rngb, B, QB, dev, 112) for _ in range(4)]
            cnt = [0]
            def s3d():
                xb, yb, xl, yl = batches[cnt[0] % 4]; cnt[0] += 1
                opt.zero_grad(); loss = sm(xb, yb, xl, yl); loss.backward(); sm.reduce_gradients(opt); opt.step()
            run("vb with bench.py's batches", s3d)
            dist.barrier()
            run("vb with bench.py's batches, after a dist.barrier()", s3d)
    if "w" in which:
        m = build(); sm = ShardedFFC(m, dist)
        optf = FusedSGD([p for p in m.parameters() if p.requires_grad], 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
        def w2():
            optf.zero_grad(); loss = sm(x, y, lab, lab); loss.backward(); sm.reduce_gradients(optf); optf.step()
        run("w2 ShardedFFC + FusedSGD (one all-reduce of the flat gradient)", w2)
        ln = lab.numpy()
        sm.exchange_labels = lambda a, b: (ln, ln)
        run("w3 = w2 without the gloo label exchange", w2)
        m.probe_net.__dict__["signal_stages"] = False
        run("w4 = w3 with the unstaged backward pass", w2)
        def w5():
            optf.zero_grad(); loss = sm(x, y, lab, lab); loss.backward(); optf.step()
        run("w5 = w4 without any gradient collective", w5)
    dist.destroy_process_group()
