#!/usr/bin/env python
"""Headline benchmark: faces/sec of the FFC training step (zero_grad -> forward -> backward ->
SGD step) on N MI355X of one node.  Metric / config: BASELINE.json `metric` — iResNet100 +
10 M identities (10 485 760 pool slots x 512, full residency: 41 GB of fp32 pool + a 10.7 GB bf16
shadow fit one 288 GB GPU), FFC DCP, bf16 MFMA operands, batch_size 256 per GPU.  `configs[1]`
(ir50 + 1M) is `--net ir50 --identities 1048576`.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement): value = whole-job faces/sec with the
inputs resident in HBM, plus `roofline` (dominant kernel family, timed live with HIP events on its
launch stream) and `cpu_baseline` (the oracle's CPU restatement on a bounded sample, rank 0, N = 1).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md chip table
PEAK_FP8_TFLOPS = 5000.0    # dense fp8 (block-scaled v_mfma_scale_*_f8f6f4 forms), same table
PMC_TRAFFIC_FILE = "r04_pmc_traffic.json"   # per-launch HBM bytes of this round's kernels (scripts/profile_round.sh / pmc_only.sh)


def kernel_source_sha():
    """Hash of csrc/ + include/ (scripts/pmc_traffic.py writes the same into the traffic file): the PMC figure is attached only
    when it was collected on exactly these kernel sources — a stale file yields traffic = null, not an old number."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(ROOT, "very-large-scale-face-recognition_amd", "csrc", "*.[hc]*")) +
                     glob.glob(os.path.join(ROOT, "include", "*.h"))):
        if os.path.isfile(fn):
            h.update(os.path.basename(fn).encode())
            h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def host_threads():
    """Cores this process may actually use (the GPU box gives one GPU's share, not the whole host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def note(msg):
    sys.stderr.write("[bench %.1fs] %s\n" % (time.perf_counter() - T0, msg))
    sys.stderr.flush()


T0 = time.perf_counter()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--net", default="ir100")
    ap.add_argument("--batch", type=int, default=256, help="--batch_size of the reference: rows of x and of y per GPU")
    ap.add_argument("--identities", type=int, default=10 << 20)
    ap.add_argument("--queue", type=int, default=0, help="pool slots (0 = one per identity, full residency)")
    ap.add_argument("--feat", type=int, default=512)
    ap.add_argument("--loss", default="Arc")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true", help="A/B: one HIP stream (no gallery / second-backward side streams)")
    ap.add_argument("--graphs", action="store_true", help="replay the backbones' forward / backward executor calls from HIP graphs "
                    "(NativeBackbone.use_graphs): for launch-bound sizes, e.g. --net mobile --feat 128 --identities 1000 --batch 32")
    ap.add_argument("--fwd-chains", type=int, default=2, choices=(2, 4), help="A/B: 2 = pass by pass, probe beside gallery (default); "
                    "4 = the four backbone passes of a step on four HIP streams (FFC.embed_both; measured slower: 93.5 vs 90.3 ms)")
    ap.add_argument("--overlap-wgrad", action="store_true", help="A/B: weight gradients of each backward pass on a side stream "
                    "(vlsfr_iresnet_backward_overlap; measured slower, off by default)")
    ap.add_argument("--counters-only", action="store_true", help="stop after the timed region: what a rocprofv3 --pmc pass needs.  On by "
                    "itself under counter collection (ROCPROF_COUNTER_COLLECTION=1): rocprofv3 --pmc over the FULL run (~36 000 "
                    "dispatches) ends in a SIGSEGV on a non-main thread inside a memcpy of the tool in about one run of three, with "
                    "the per-launch event brackets taken from a pre-created pool as well as with per-launch hipEventCreate "
                    "(gpurun_out/pmc_full, prof_r03): a tool-side failure that grows with the dispatch count, so PMC passes stay short")
    ap.add_argument("--full-under-pmc", action="store_true", help="do NOT switch to --counters-only under rocprofv3 counter collection "
                    "(scripts/pmc_full_once.sh: the one full-length pass that reproduces / attributes the round-3 fault)")
    ap.add_argument("--sync-debug", action="store_true", help="diagnostic: torch.cuda.set_sync_debug_mode('warn') around two steps")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-GPU code path (process group, identity-sharded pool, "
                    "partitioned SGD, every collective) even with one rank: rehearses the RCCL calls on a 1-GPU box")
    ap.add_argument("--rehearse-world", type=int, default=0, help="ONE process runs rank 0's step of a W-rank job (shard-local pool "
                    "of identities / W slots, gathered batch of W x batch rows, replicated LRU, partitioned SGD) with every collective "
                    "replaced by a local stand-in of the same shape: rank 0's compute at the node's shapes, no wire time "
                    "(parallel.RehearsalDist).  Config C4: --rehearse-world 8 --identities 104857600 --batch 64")
    ap.add_argument("--head-dtype", default="bf16", choices=["bf16", "fp8"], help="fp8: the e4m3 sweep of csrc/head8.hip (config C5's "
                    "precision for the class matmul; the backbone stays bf16)")
    ap.add_argument("--cu-reserve", type=int, default=0, help="experiment: run the probe chain and the gallery / second-backward chain on CU-masked "
                    "streams that each leave this many CUs (a multiple of 8) to the other chain")
    ap.add_argument("--phases", action="store_true", help="diagnostic: print the forward / backward / update split to stderr")
    ap.add_argument("--timed-profile", action="store_true", help="per-launch HIP events inside the timed region itself (they cost "
                    "~1.5 %% of the step: by default the K timed steps run clean and are REPEATED with the events on)")
    ap.add_argument("--opt", action="append", default=[], help="name=value tuning switch (vlsfr_set_option), repeatable")
    ap.add_argument("--conv-glds", type=int, default=-1, help="A/B switch for the conv kernel variant (vlsfr_set_option)")
    ap.add_argument("--pool", default="sharded", choices=["sharded", "replicated"],
                    help="N > 1: identity-sharded pool (softmax all-reduce) or replicated pool")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI); gloo only to rehearse the "
                    "multi-process path on a 1-GPU box")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--c1-steps", type=int, default=5, help="timed steps of the C1 leg of the CPU baseline (0 = skip)")
    return ap.parse_args()


def synth_batch(rng, B, n_id, device, hw=112):
    """uniform uint8 pixels -> (v - 127.5) * 0.0078125 fp32 NCHW; id half shares labels
    between the two views, instance half draws them independently (main.py:53-60)."""
    def imgs():
        u8 = torch.from_numpy(rng.integers(0, 256, size=(B, 3, hw, hw), dtype=np.uint8)).to(device)
        return (u8.float() - 127.5) * 0.0078125
    h = B // 2
    ids = rng.choice(n_id, size=h, replace=False)
    xl = np.concatenate([ids, rng.integers(0, n_id, size=B - h)]).astype(np.int64)
    yl = np.concatenate([ids, rng.integers(0, n_id, size=B - h)]).astype(np.int64)
    return imgs(), imgs(), torch.from_numpy(xl), torch.from_numpy(yl)


def _cpu_steps(o, B, n_id, steps, tag, hw=112):
    """Times `steps` oracle steps (zero_grad -> forward -> backward -> SGD-nesterov) after one warm-up step."""
    from oracle import ffc_ref
    rng = np.random.default_rng(0)
    bufs = [None] * len(o.parameters())
    t_total, faces = 0.0, 0
    for step in range(steps + 1):
        x, y, xl, yl = synth_batch(rng, B, n_id, "cpu", hw)
        note("cpu baseline %s step %d" % (tag, step))
        t0 = time.perf_counter()
        for p in o.parameters():
            p.grad = None
        loss = o.forward(x, y, xl, yl)
        loss.backward()
        ps = o.parameters()
        bufs = ffc_ref.sgd_nesterov_step_ref(ps, [p.grad for p in ps], bufs, 0.1)
        dt = time.perf_counter() - t0
        if step > 0:                                      # first step = warm-up
            t_total += dt
            faces += 2 * B
    return faces / t_total


def cpu_baseline(args):
    """The oracle's CPU restatement (PyTorch CPU fp32, all host cores of this process's share) on two bounded samples:
    (i) `value`: the GPU line's own backbone and feature size at a reduced batch and pool (the 10 M-slot pool and
    batch 256 would take minutes per step on the host), (ii) `c1`: BASELINE.json configs[0] EXACTLY — MobileFaceNet,
    D = 128, 1 000 identities / slots, batch_size 32, fp32 (the imported reference ran this at
    ~42 faces/s on the build container's 8 vCPUs)."""
    from oracle import ffc_ref
    threads = host_threads()
    torch.set_num_threads(threads)
    B, Q = args.cpu_batch, 1 << 17
    gen = torch.Generator().manual_seed(0)
    o = ffc_ref.FFCRef(args.net, args.feat, Q, 32.0, args.loss, 0.5, 0.99, gen=gen)
    o.lru.restore([(k, k) for k in range(4096)])          # a few thousand resident identities (O(n) oracle LRU)
    value = _cpu_steps(o, B, 4096, args.cpu_steps, args.net, 224 if args.net in ("r50", "r101") else (64 if args.net == "rtiny" else 112))
    out = dict(value=value, unit="faces/sec", cores=threads, kind="port",
               sample="%s D=%d, pool 131072 slots, batch_size %d, %d timed steps after 1 warm-up, fp32 PyTorch-CPU oracle" %
                      (args.net, args.feat, B, args.cpu_steps))
    if args.c1_steps > 0:
        gen = torch.Generator().manual_seed(0)
        o1 = ffc_ref.FFCRef("mobile", 128, 1000, 32.0, "Arc", 0.5, 0.99, gen=gen)
        out["c1"] = dict(value=_cpu_steps(o1, 32, 1000, args.c1_steps, "C1"), unit="faces/sec", cores=threads,
                         sample="BASELINE configs[0] exactly: MobileFaceNet D=128, 1000 identities, pool 1000 slots, "
                                "batch_size 32, Arc, fp32, %d timed steps after 1 warm-up" % args.c1_steps)
    return out


_RESULT_FD = None


def emit(line):
    """The one JSON line of the contract, on the process's ORIGINAL stdout."""
    if _RESULT_FD is None:
        print(line, flush=True)
    else:
        os.write(_RESULT_FD, (line + "\n").encode())


def main():
    global _RESULT_FD
    args = parse()
    # stdout carries exactly one line (rank 0's JSON): libraries that print to fd 1 on their own — RCCL's version banner at
    # communicator creation, gloo's connection notes — go to stderr with everything else
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)
    _skip = os.environ.get("BENCH_SKIP", "").split(",")      # diagnostic A/B switches
    if "threads" not in _skip:
        torch.set_num_threads(host_threads())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 bench.py --gpus %d ..." % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    rehearse = args.rehearse_world if args.rehearse_world > 1 else 0
    if rehearse and world != 1:
        raise SystemExit("--rehearse-world runs in one process")
    dist_on = world > 1 or args.force_dist or bool(rehearse)   # --force-dist: the N > 1 code path (RCCL collectives included) with one rank
    if rehearse:
        from vlsfr_amd.parallel import RehearsalDist
        dist = RehearsalDist(rehearse, args.identities, seed=99)
    elif dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # torch's stream pool comes to life BEFORE RCCL creates its streams: HIP deals its few hardware queues to streams in
        # creation order, and with the communicator's streams first the step's side streams (gallery pass, second backward
        # pass) landed on queues they had to share — 101 ms per step instead of 92.6 with one rank (BENCH_SKIP=late_streams
        # restores that order for an A/B).
        from vlsfr_amd.parallel import warm_stream_pool
        if "late_streams" not in _skip:
            warm_stream_pool(dev)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    import vlsfr_amd  # noqa: F401
    from vlsfr_amd import _lib
    from vlsfr_amd.ffc import FFC
    from vlsfr_amd.optim import get_optim_scheduler
    from vlsfr_amd.parallel import DataParallelFFC, ShardedFFC

    Q = args.queue or args.identities
    torch.manual_seed(1234)                                   # identical initial weights on every rank
    pool_world = rehearse or world                             # ranks the pool is split over
    sharded = dist_on and args.pool == "sharded" and Q % pool_world == 0
    # the pool is drawn straight into HBM in chunks (ffc.build_pool: same normalize(rand) semantics as
    # ffc.py:29-30); under the identity-sharded pool every rank builds only its own slots
    model = FFC(args.net, args.feat, Q, 32.0, args.loss, 0.5, 0.99, pool_device=dev,
                pool_shard=(rank, pool_world) if sharded else None).cuda()
    model.__dict__['head_dtype'] = args.head_dtype
    model.__dict__['forward_chains'] = args.fwd_chains

    def set_graphs(on):
        for net in (model.probe_net, model.gallery_net):
            if hasattr(net, "use_graphs"):
                net.use_graphs = bool(on)
    set_graphs(args.graphs)
    if args.cu_reserve:
        from vlsfr_amd.parallel import cu_masked_stream
        main_m = cu_masked_stream(dev, args.cu_reserve, low=True)
        side_m = cu_masked_stream(dev, args.cu_reserve, low=False)
        main_m.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(main_m)
        model.__dict__['_side_stream'] = side_m
        model.probe_net._bwd_stream = side_m
    if args.serial:
        model.__dict__['concurrent_streams'] = False
        model.probe_net.concurrent_backward = False
        model.probe_net.overlap_wgrad = False
    elif args.overlap_wgrad:
        model.probe_net.overlap_wgrad = True
    n_res = min(Q, args.identities)
    ar = np.arange(n_res)
    model.lru.restore_arrays(ar.astype(np.int64), ar.astype(np.int32))   # steady state: the pool is full (lru.py:113)
    note("pool %d slots on device (%.1f GB), LRU restored" % (Q, model.queue.numel() * 4 / 1e9))
    cfg = dict(optim="SGD", scheduler="multistep", LR=0.1, momentum=0.9, decay=1e-4, nesterov=True, warmup=0, epochs=1,
               milestones=[8, 14, 17], gammas=[0.1, 0.1, 0.1])     # config/optim_config
    step_model = model
    if dist_on:
        # N > 1: partitioned SGD (each rank updates 1/N of the parameters; gradients reduce-scattered bucket by bucket
        # under the backward pass), behind the reference's scheduler interface
        from vlsfr_amd.optim.optimizer import WarmupSchedule
        step_model = ShardedFFC(model, dist) if sharded else DataParallelFFC(model, dist)
        opt = step_model.make_optimizer(cfg["LR"], cfg["momentum"], cfg["decay"], cfg["nesterov"])
        sched = WarmupSchedule(opt, "multistep", cfg["warmup"], cfg["epochs"], milestones=cfg["milestones"], gammas=cfg["gammas"])
    else:
        opt, sched = get_optim_scheduler([p for p in model.parameters() if p.requires_grad], cfg)
    sched.update(0, 0.0)
    rng = np.random.default_rng(1234 + rank)
    B = args.batch
    hw = model.probe_net.image_size                          # 112 (iResNet / MobileFaceNet), 224 for r50
    batches = [synth_batch(rng, B, args.identities, dev, hw) for _ in range(min(4, args.steps + args.warmup))]

    phase_ev = []

    def one_step(i, phases=False):
        x, y, xl, yl = batches[i % len(batches)]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if phases else None
        if phases:
            ev[0].record()
        opt.zero_grad()
        loss = step_model(x, y, xl, yl)
        if phases:
            ev[1].record()
        loss.backward()
        if dist_on:
            step_model.reduce_gradients(opt)
        if phases:
            ev[2].record()
        opt.step()
        if phases:
            ev[3].record()
            phase_ev.append(ev)
        return loss

    L = _lib.lib()
    if args.conv_glds >= 0:
        L.vlsfr_set_option(b"conv_glds", ctypes.c_int32(args.conv_glds))
    for kv in args.opt:
        k, v = kv.split("=")
        _lib.check(L.vlsfr_set_option(k.encode(), ctypes.c_int32(int(v))), "vlsfr_set_option")
    note("model on device, pool %d slots; warm-up" % Q)
    for i in range(args.warmup):
        one_step(i)
        torch.cuda.synchronize()
        note("warm-up step %d done" % i)
    if dist is not None:
        dist.barrier()
    if rehearse:
        torch.cuda.synchronize()
        note("rehearsal of rank 0 of %d: HBM in use %.1f GB (pool shard %.1f GB fp32 + its %s shadow), host RSS %.1f GB" %
             (rehearse, torch.cuda.memory_allocated() / 1e9, model.queue.numel() * 4 / 1e9, args.head_dtype,
              __import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1e6))
    if args.sync_debug:
        torch.cuda.set_sync_debug_mode("warn")
        for i in range(2):
            one_step(args.warmup + i)
        torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
    if args.phases:          # diagnostic: forward / backward / update split of the three-stream schedule (main-stream events)
        step_model.__dict__["_marks"] = []
        for i in range(4):
            one_step(args.warmup + i, phases=True)
        torch.cuda.synchronize()
        marks = step_model.__dict__.pop("_marks", None) if hasattr(step_model, "__dict__") else None
        if marks:
            per = len(marks) // 4
            for k in range(1, per):
                note("  forward segment %-44s %.2f ms" % (marks[k][0], sum(marks[s * per + k - 1][1].elapsed_time(marks[s * per + k][1])
                                                                          for s in range(1, 4)) / 3))
        for k, nm in enumerate(("forward (2 passes + heads)", "backward (2 passes)", "optimizer + EMA")):
            note("phase %-28s %.2f ms" % (nm, sum(e[k].elapsed_time(e[k + 1]) for e in phase_ev[1:]) / (len(phase_ev) - 1)))
    L.vlsfr_profile_reset()
    L.vlsfr_profile_enable(1 if args.timed_profile else 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_ms = []
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    evs[0].record()
    for i in range(args.steps):
        th = time.perf_counter()
        loss = one_step(args.warmup + i)
        evs[i + 1].record()
        host_ms.append((time.perf_counter() - th) * 1e3)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L.vlsfr_profile_enable(0)
    note("timed region done: %.3f s for %d steps; GPU ms per step: %s; host-side issue ms per step: %s" %
         (dt, args.steps, " ".join("%.1f" % evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)),
          " ".join("%.1f" % v for v in host_ms)))
    if dist is not None and not rehearse:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    def collect():
        out = {}
        # family 3: the input-gradient launches whose epilogue also accumulates the BatchNorm-backward reduction (a separate
        # kernel instantiation doing more than the convolution: priced on the convolution's FLOPs only, reported apart)
        for fam, name in ((0, "conv_igemm_kernel"), (1, "conv_wgrad_kernel"), (2, "head_sweep_kernel"), (3, "conv_igemm_bnred_kernel")):
            ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
            L.vlsfr_profile_collect.restype = ctypes.c_int
            _lib.check(L.vlsfr_profile_collect(ctypes.c_int32(fam), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)))
            out[name] = (ms.value, fl.value, n.value)
        L.vlsfr_profile_dropped.restype = ctypes.c_int64
        dropped = int(L.vlsfr_profile_dropped())
        if dropped:
            note("WARNING: %d launch brackets were dropped (event pool exhausted): the roofline totals undercount" % dropped)
        L.vlsfr_profile_reset()
        return out

    if os.environ.get("ROCPROF_COUNTER_COLLECTION") == "1" and rank == 0:
        # the executable mappings of this process, once: an unsymbolised stack trace of a later fault (the rocprofv3 --pmc pass of
        # round 3 died on a thread of libhsa-runtime64.so: scripts/match_frames.py) is attributable from them
        try:
            with open("/proc/self/maps") as f:
                for ln in f:
                    if " r-xp " in ln and (".so" in ln or "python" in ln):
                        sys.stderr.write("[maps] " + ln)
            sys.stderr.flush()
        except OSError:
            pass
    if os.environ.get("ROCPROF_COUNTER_COLLECTION") == "1" and not args.counters_only and not args.full_under_pmc:
        note("rocprofv3 counter collection detected (ROCPROF_COUNTERS=%s): stopping after the timed region (--counters-only)" %
             os.environ.get("ROCPROF_COUNTERS", "?"))
        args.counters_only = True
    if args.counters_only:   # rocprofv3 --pmc passes: the K timed steps are all that is wanted (no HIP events anywhere)
        if rank == 0:
            emit(json.dumps({"value": round(world * 2 * B * args.steps / dt, 2), "unit": "faces/sec", "ms_per_step": round(dt / args.steps * 1e3, 3),
                             "note": "--counters-only run: no roofline legs"}))
        return
    set_graphs(False)      # the per-launch event brackets below need plain launches
    L.vlsfr_profile_event_overhead_us.restype = ctypes.c_double
    ev_us = float(L.vlsfr_profile_event_overhead_us(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    if not args.timed_profile:
        # `value` comes from K clean steps; the same schedule is repeated with a HIP-event bracket around every launch of
        # the three contraction families on its own stream (roofline.timed_region)
        L.vlsfr_profile_enable(1)
        for i in range(min(args.steps, 5)):
            one_step(args.warmup + i)
        torch.cuda.synchronize()
        L.vlsfr_profile_enable(0)
    fams_timed = collect()
    # Per-kernel rate: in the timed region two or three HIP streams share the CUs, so a launch's event
    # duration includes the time it shared the chip with another kernel.  The roofline figure is taken
    # from a serialized replay of the same steps (one stream, same kernels, same shapes) right after the
    # timed region; the timed-region figure is reported next to it.
    serial_steps = 0 if args.serial else min(3, args.steps)
    fams = fams_timed
    if serial_steps:
        model.__dict__['concurrent_streams'] = False
        model.probe_net.concurrent_backward = False
        model.probe_net.overlap_wgrad = False
        one_step(0)
        torch.cuda.synchronize()
        L.vlsfr_profile_enable(1)
        for i in range(serial_steps):
            one_step(1 + i)
        torch.cuda.synchronize()
        L.vlsfr_profile_enable(0)
        fams = collect()
        model.__dict__['concurrent_streams'] = True
        model.probe_net.concurrent_backward = True
        model.probe_net.overlap_wgrad = args.overlap_wgrad
    loss_val = float(step_model.global_loss(loss)) if dist_on else float(loss.detach())   # collective: every rank
    if rank != 0:
        return
    # an event bracket reads the kernel plus the empty-bracket time measured above: take that out per launch
    debracket = lambda v: (max(v[0] - v[2] * ev_us * 1e-3, 1e-9), v[1], v[2])
    fams = {k: debracket(v) for k, v in fams.items()}
    fams_timed = {k: debracket(v) for k, v in fams_timed.items()}
    # The convolution family is priced WHOLE: the input-gradient launches that also accumulate the BatchNorm-backward reduction
    # (conv_igemm_bnred_kernel, event family 3) are the same convolutions with a longer epilogue, so their FLOPs and their time
    # are folded into conv_igemm_kernel for the headline figure (leaving the slowest tenth of the family's FLOPs
    # out flattered `frac`); the split stays visible in roofline.conv_split.
    conv_split = None
    def fold(d):
        a, b = d["conv_igemm_kernel"], d.pop("conv_igemm_bnred_kernel")
        d["conv_igemm_kernel"] = (a[0] + b[0], a[1] + b[1], a[2] + b[2])
        return a, b
    plain, red = fold(fams)
    fold(fams_timed)
    tf = lambda v: round(v[1] / (v[0] * 1e-3) / 1e12, 2) if v[0] > 0 else 0.0
    conv_split = dict(plain=dict(total_ms=round(plain[0], 2), tflops=tf(plain), launches=int(plain[2])),
                      with_bn_backward_reduction=dict(total_ms=round(red[0], 2), tflops=tf(red), launches=int(red[2])),
                      frac_without_bnred_launches=round(tf(plain) / PEAK_BF16_TFLOPS, 4))
    dom = max(fams, key=lambda k: fams[k][0])
    ms, fl, n = fams[dom]
    achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    tms, tfl, tn = fams_timed[dom]
    head_peak = PEAK_FP8_TFLOPS if args.head_dtype == "fp8" else PEAK_BF16_TFLOPS      # the sweep's MFMA form
    dom_peak = head_peak if dom == "head_sweep_kernel" else PEAK_BF16_TFLOPS
    roofline = dict(bound="mfma", kernel=dom, achieved=round(achieved, 2), peak=dom_peak, unit="TFLOP/s",
                    frac=round(achieved / dom_peak, 4), traffic=None, launches=int(n),
                    avg_launch_us=round(ms * 1e3 / max(n, 1), 2), event_bracket_us_subtracted=round(ev_us, 2),
                    measured_in=("timed region (single stream)" if not serial_steps else
                                 "serialized replay of %d steps after the timed region (one stream)" % serial_steps),
                    timed_region=None if tn == 0 else dict(
                        achieved=round(tfl / (tms * 1e-3) / 1e12, 2) if tms > 0 else 0.0,
                        avg_launch_us=round(tms * 1e3 / max(tn, 1), 2), launches=int(tn),
                        note="the timed schedule repeated with HIP events on every launch; streams overlap: launch durations include shared-chip time"),
                    other={k: dict(total_ms=round(v[0], 2), tflops=round(v[1] / (v[0] * 1e-3) / 1e12, 2) if v[0] > 0 else 0,
                                   launches=int(v[2])) for k, v in fams.items() if k != dom})
    if dom == "conv_igemm_kernel":
        roofline["conv_split"] = conv_split
        roofline["family"] = "conv_igemm: every forward and input-gradient convolution launch, the bnred launches included"
    else:
        roofline["other"]["conv_igemm_kernel"]["conv_split"] = conv_split
    hs = fams.get("head_sweep_kernel")
    if hs and hs[2] > 0 and hs[0] > 0:
        # the class matmul north_star singles out: MFMA fraction of both contractions (4 B Q D FLOPs per sweep) and
        # the HBM rate of the pool bytes it streams (bf16 shadow: Q * D * 2 per sweep)
        pool_bytes = Q // max(pool_world if sharded else 1, 1) * args.feat * 2
        (roofline if dom == "head_sweep_kernel" else roofline["other"]["head_sweep_kernel"]).update(
            avg_launch_us=round(hs[0] * 1e3 / hs[2], 1), mfma_frac=round(hs[1] / (hs[0] * 1e-3) / 1e12 / head_peak, 4), mfma_peak=head_peak,
            pool_gb_per_s=round(pool_bytes * hs[2] / (hs[0] * 1e-3) / 1e9, 1))
    # HBM traffic per launch of the dominant kernel: PMC counters cannot be collected from inside this
    # process, so the figure measured by rocprofv3 --pmc on this same command (profiles/) is attached
    # when the configuration matches
    try:
        with open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)) as f:
            pt = json.load(f)
        if pt["config"] == {"net": args.net, "batch": B, "identities": args.identities}:
            if pt.get("source_sha") == kernel_source_sha():
                roofline["traffic"] = pt["kernels"][dom.replace("_kernel", "")]["hbm_bytes_per_launch"]
                roofline["traffic_source"] = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes of this command, "
                                              "collected on these kernel sources: sha %s)" % (PMC_TRAFFIC_FILE, pt["source_sha"]))
            else:
                roofline["traffic_source"] = "profiles/%s is from other kernel sources (sha %s, here %s): not attached" % (
                    PMC_TRAFFIC_FILE, pt.get("source_sha"), kernel_source_sha())
    except (OSError, KeyError, ValueError):
        pass
    faces = world * 2 * B * args.steps
    out = {
        "metric": ("faces/sec (whole node) at 10M-identity FFC, iResNet100, 1/2/4/8 MI355X"
                   if (args.net, args.identities) == ("ir100", 10 << 20) else
                   "faces/sec (whole node) at %s-identity FFC, %s" % (
                       "%dM" % (args.identities >> 20) if args.identities >= (1 << 20) else str(args.identities), args.net)),
        "value": round(faces / dt, 2), "unit": "faces/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.head_dtype == "bf16" else "bf16 backbone + fp8 (e4m3) class matmul", "data": "synthetic",
        "config": {"workload": "%s + %d identities, FFC DCP (pool %d slots x %d, loss %s), batch_size %d per GPU "
                               "(2 x %d faces per step per GPU), SGD-nesterov, %dx%d synthetic images" %
                               (args.net, args.identities, Q, args.feat, args.loss, B, B, hw, hw),
                   "parallelism": ("dp%d" % world) + ("" if not dist_on else "+zero1-sgd+pool-" + ("sharded" if isinstance(step_model, ShardedFFC) else "replicated")) +
                                  ("" if not rehearse else " REHEARSAL: rank 0 of %d in one process, collectives replaced by local stand-ins of the same "
                                   "shape (no wire time); value = this GPU's faces/sec, gathered batch %d rows, pool shard %d slots" % (rehearse, rehearse * B, Q // rehearse)),
                   "loss": loss_val},
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    emit(json.dumps(out))


if __name__ == "__main__":
    main()
