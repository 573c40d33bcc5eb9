"""The whole N > 1 step rehearsed with TWO ranks on ONE GPU (torch.distributed over gloo: the collectives stage through
the host; everything else — kernels, streams, per-bucket events, the partitioned optimizer, the identity-sharded head —
is the code the RCCL run uses).  RCCL itself only runs in the driver's multi-GPU bench: no multi-GPU box is available
to the build, and this file says so rather than pretending otherwise.

Checked: (1) step-1 loss of the 2-rank run == a single-process evaluation of the same global batch with per-rank
BatchNorm (the semantics parallel.py defines), (2) identity-sharded pool + partitioned SGD == replicated pool +
all-reduce + replicated SGD after two steps (losses, parameters, pool rows, LRU / queue_position state)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NET, D, Q, B, WORLD, STEPS = "irtiny", 64, 256, 8, 2, 2


def _batches(rank):
    from tests.golden import common
    rng = np.random.default_rng(500 + rank)
    out = []
    for _ in range(STEPS):
        ids = rng.choice(300, size=B // 2, replace=False)
        xl = np.concatenate([ids, rng.integers(0, 300, B - B // 2)]).astype(np.int64)
        yl = np.concatenate([ids, rng.integers(0, 300, B - B // 2)]).astype(np.int64)
        out.append((common.images_from_u8(common.synth_images_u8(rng, B)), common.images_from_u8(common.synth_images_u8(rng, B)),
                    torch.from_numpy(xl), torch.from_numpy(yl)))
    return out


def _build():
    from vlsfr_amd.ffc import FFC
    torch.manual_seed(0)
    return FFC(NET, D, Q, 32.0, "Arc", 0.5, 0.99, precise_head=True).cuda()


def _worker(rank, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # a stuck rank dumps its stacks (and the run is bounded by the parent's queue timeout) instead of hanging silently
    import faulthandler
    logdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(logdir, exist_ok=True)
    trace = open(os.path.join(logdir, "two_rank_worker%d.log" % rank), "w")
    faulthandler.dump_traceback_later(150, repeat=False, file=trace, exit=True)

    def note(msg):
        trace.write(msg + "\n")
        trace.flush()
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        import vlsfr_amd  # noqa: F401
        from vlsfr_amd.optim.fused import FusedSGD
        from vlsfr_amd.parallel import DataParallelFFC, ShardedFFC
        res = {}
        for mode in ("sharded", "replicated"):
            m = _build()
            if mode == "sharded":
                sm = ShardedFFC(m, dist)
                opt = sm.make_optimizer(0.1, 0.9, 1e-4, True)
            else:
                sm = DataParallelFFC(m, dist)
                opt = FusedSGD([p for p in m.parameters() if p.requires_grad], 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
            losses = []
            note("%s: built" % mode)
            for x, y, xl, yl in _batches(rank):
                opt.zero_grad()
                loss = sm(x.cuda(), y.cuda(), xl, yl)
                note("%s: forward enqueued" % mode)
                loss.backward()
                sm.reduce_gradients(opt)
                note("%s: gradients reduced" % mode)
                opt.step()
                losses.append(float(sm.global_loss(loss)))
                note("%s: step done, loss %.5f" % (mode, losses[-1]))
            torch.cuda.synchronize()
            pool = sm.gather_pool() if mode == "sharded" else m.queue
            if mode == "sharded":
                st = sm.pool_state()                       # shard-wise checkpoint round trip
                sm.load_pool_state(st)
                assert torch.equal(sm.gather_pool(), pool)
            # numpy arrays travel through the queue by value (tensors would be shared-memory handles of a process
            # that may be gone when the parent reads them)
            res[mode] = dict(losses=losses, w=torch.cat([p.detach().float().reshape(-1).cpu() for p in m.probe_net.parameters()]).numpy(),
                             pool=pool.cpu().numpy(), lru=m.lru.state_dict(), qp=m._state().qp.tolist())
        out.put((rank, res))
        note("results queued")
        faulthandler.cancel_dump_traceback_later()
    finally:
        dist.destroy_process_group()


def test_two_rank_step_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, out)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=240) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # (1) single-process evaluation of the global batch with per-rank BatchNorm
    m = _build()
    head = m._ensure_head()
    data = [_batches(r)[0] for r in range(WORLD)]
    xl = np.concatenate([d[2].numpy() for d in data])
    yl = np.concatenate([d[3].numpy() for d in data])
    with torch.no_grad():
        m._momentum_update_gallery()
        p1 = torch.cat([m.probe_net(d[0].cuda()) for d in data])
        g1 = torch.cat([m.gallery_net(d[1].cuda()) for d in data])
        loss2 = head.run_pass(p1, g1, xl, yl, True)
        p2 = torch.cat([m.probe_net(d[1].cuda()) for d in data])
        g2 = torch.cat([m.gallery_net(d[0].cuda()) for d in data])
        loss1 = head.run_pass(p2, g2, yl, xl, False)
    want = float(loss1 + loss2)
    for mode in ("sharded", "replicated"):
        for r in range(WORLD):
            np.testing.assert_allclose(res[r][mode]["losses"][0], want, rtol=2e-3)
    # (2) sharded pool + partitioned SGD == replicated pool + all-reduce (and both ranks agree with each other)
    a, b = res[0]["sharded"], res[0]["replicated"]
    rel = lambda u, v: float(np.linalg.norm(u - v) / (np.linalg.norm(v) + 1e-30))
    # step 1: same weights, same batch -> only kernel-level noise; step 2 follows an lr = 0.1 update whose gradients carry the
    # atomic-order noise of a batch-8 train-mode-BN net (run-to-run spread of the same mode: ~1 %)
    np.testing.assert_allclose(a["losses"][0], b["losses"][0], rtol=3e-3)
    np.testing.assert_allclose(a["losses"], b["losses"], rtol=2e-2)
    assert rel(a["w"], b["w"]) < 2e-2, rel(a["w"], b["w"])
    assert a["lru"] == b["lru"] and a["qp"] == b["qp"]
    changed = np.abs(a["pool"] - b["pool"]).max(axis=2) > 0
    if changed.any():
        cos = torch.nn.functional.cosine_similarity(torch.from_numpy(a["pool"][changed]), torch.from_numpy(b["pool"][changed]), dim=1)
        assert float(cos.min()) > 0.995
    for mode in ("sharded", "replicated"):
        assert rel(res[1][mode]["w"], res[0][mode]["w"]) < 1e-6          # parameters stay replicated
        assert res[1][mode]["lru"] == res[0][mode]["lru"] and res[1][mode]["qp"] == res[0][mode]["qp"]
        assert np.array_equal(res[1][mode]["pool"], res[0][mode]["pool"])


def _train_worker(rank, port, out, tmp, pool="sharded"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK="0")
    import faulthandler
    logdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(logdir, exist_ok=True)
    trace = open(os.path.join(logdir, "two_rank_train%d.log" % rank), "w")
    faulthandler.dump_traceback_later(200, repeat=False, file=trace, exit=True)
    try:
        from vlsfr_amd.main import parse_args, train
        base = ["--net_type", "irtiny", "--feat_dim", "32", "--queue_size", "64", "--batch_size", "8", "--print_freq", "2",
                "--iters_per_epoch", "4", "--num_class", "500", "--dist_backend", "gloo", "--pool", pool]
        res = {}
        for tag, extra in (("a", []), ("b", ["--resume", os.path.join(tmp, "a", "1.pt")])):
            net, loss = train(parse_args(base + ["--saved_dir", os.path.join(tmp, tag)] + extra), log=lambda *_: None)
            torch.cuda.synchronize()
            trace.write("run %s done\n" % tag)
            trace.flush()
            res[tag] = dict(loss=float(loss.detach()), lru=net.lru.state_dict(), qp=net._state().qp.tolist(),
                            shard=net.queue.cpu().numpy(), w=net.probe_net.state_dict()["layer1.0.conv1.weight"].float().cpu().numpy())
        res["files"] = sorted(os.listdir(os.path.join(tmp, "a")))
        out.put((rank, res))
        faulthandler.cancel_dump_traceback_later()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("pool", ["sharded", "replicated"])
def test_two_rank_training_driver_checkpoints_and_resumes(tmp_path, pool):
    """main.train with WORLD_SIZE = 2 (gloo, one GPU), ZeRO-1 optimizer, checkpoint every 2 of 4 iterations, `--resume`
    from the first one continues to the state of the uninterrupted run — including the optimizer's momenta, which live as
    1 / world slices and must be gathered for the file on every rank, whichever pool form runs (the update made AFTER the
    resume point is compared: a run that lost its momenta makes a ~45 % different one).
    pool "sharded": pool built shard-local; files: rank 0 the model (fc / lru / qp = None), every rank its pool slots and the
    replicated allocator state as arrays.  pool "replicated": one file in the reference's format (main.py:85)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, port, out, str(tmp_path), pool)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=300) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ck = torch.load(os.path.join(str(tmp_path), "a", "1.pt"), weights_only=True)
    if pool == "sharded":
        assert res[0]["files"] == ["1.pool0.pt", "1.pool1.pt", "1.pt", "2.pool0.pt", "2.pool1.pt", "2.pt"]
        assert ck["fc"] is None and ck["lru"] is None and ck["qp"] is None
        ps = torch.load(os.path.join(str(tmp_path), "a", "1.pool1.pt"), weights_only=True)
        assert ps["fc_shard"].shape == (2, 32, 32) and ps["qp"].shape == (64,) and ps["lru_keys"].shape == ps["lru_slots"].shape
    else:
        assert res[0]["files"] == ["1.pt", "2.pt"]
        assert ck["fc"].shape == (2, 64, 32) and len(ck["qp"]) == 64
    mom = [v.get("momentum_buffer") for v in ck["resume"]["optimizer"]["state"].values()]
    assert len(mom) > 10 and all(m is not None and float(m.float().abs().max()) > 0 for m in mom)     # the momenta are IN the file
    w_ck = ck["state_dict"]["layer1.0.conv1.weight"].float().cpu().numpy()
    rel = lambda u, v: float(np.linalg.norm(u - v) / (np.linalg.norm(v) + 1e-30))
    for r in range(WORLD):
        a, b = res[r]["a"], res[r]["b"]
        assert a["shard"].shape == ((2, 32, 32) if pool == "sharded" else (2, 64, 32))
        assert a["lru"] == b["lru"] and a["qp"] == b["qp"]
        assert abs(a["loss"] - b["loss"]) <= 2e-2 * abs(a["loss"])
        assert float(np.abs(a["shard"] - b["shard"]).max()) < 0.05
        assert rel(b["w"], a["w"]) < 2e-2
        assert rel(b["w"] - w_ck, a["w"] - w_ck) < 0.2, rel(b["w"] - w_ck, a["w"] - w_ck)
    assert rel(res[1]["a"]["w"], res[0]["a"]["w"]) < 1e-6


def test_bench_multi_gpu_code_path_over_rccl_with_one_rank():
    """bench.py --force-dist: process group on the nccl (= RCCL) backend, stream pool warmed before it, shard-local pool,
    every collective of the sharded head and of the partitioned SGD through RCCL, per-stage events — with the one rank a
    1-GPU box has.  stdout must carry exactly the one JSON line of the contract."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--net", "irtiny", "--feat", "512", "--identities", "8192",
                        "--batch", "16", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    out = json.loads(lines[0])
    assert out["config"]["parallelism"] == "dp1+zero1-sgd+pool-sharded" and out["n_gpus"] == 1
    assert out["value"] > 0 and np.isfinite(out["config"]["loss"])
    assert out["roofline"]["achieved"] > 0
