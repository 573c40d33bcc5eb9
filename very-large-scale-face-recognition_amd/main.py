"""Training driver with the loop shape and CLI of the reference's main.py (main.py:23-86
train_one_epoch, :88-143 train, :146-170 argparse), on the MI355X path.

Differences, all forced by scope: the LMDB loaders of the reference do not import
(`util/lmdb_loader.py:4`), so batches come from `SyntheticFaces`, which honours the loader's output
contract (float32 CHW, (v - 127.5) * 0.0078125, labels; pair dataset -> (img1, img2, id)); no
autocast / GradScaler (bf16 MFMA operands with fp32 accumulation need no loss scaling); one process per
GPU under torch.distributed when WORLD_SIZE > 1 (parallel.py).  Checkpoints keep the reference's
dictionary (`main.py:85`): state_dict / lru / fc / qp.
"""
import argparse
import os
import random
import time

import numpy as np
import torch

from .ffc import FFC
from .optim import get_optim_scheduler

OPTIM_CONFIG = dict(scheduler="multistep", epochs=1, warmup=0, patience=4, milestones=[8, 14, 17],
                    gammas=[0.1, 0.1, 0.1], LR_min=1e-5, optim="SGD", LR=0.1, decay=1e-4, momentum=0.9,
                    nesterov=True)     # config/optim_config:1-14


def load_config(path):
    """Typed-JSON loader with the reference's format ["type", value] (util/config.py:4-43)."""
    import json
    conv = dict(int=int, float=float, str=str, bool=lambda v: bool(int(v)), none=lambda v: None)
    with open(path) as f:
        raw = json.load(f)
    out = {}
    for k, (t, v) in raw.items():
        out[k] = [conv[t](x) for x in v] if isinstance(v, list) else conv[t](v)
    return out


class SyntheticFaces(object):
    """Stand-in for MultiLMDBDataset + PairLMDBDataset (util/lmdb_loader.py:12-237): `n_batches`
    batches of uniform-uint8 images normalised as the loader does, instance labels uniform in
    [0, num_class), pair ids drawn without replacement."""

    def __init__(self, num_class, batch_size, n_batches, device, seed=0, image_size=112):
        self.num_class, self.B, self.n, self.dev, self.hw = num_class, batch_size, n_batches, device, image_size
        self.rng = np.random.default_rng(seed)

    def __len__(self):
        return self.n

    def _images(self, n):
        u8 = torch.from_numpy(self.rng.integers(0, 256, size=(n, 3, self.hw, self.hw), dtype=np.uint8))
        return ((u8.to(self.dev, non_blocking=True).float() - 127.5) * 0.0078125)

    def __iter__(self):
        h = self.B // 2
        for _ in range(self.n):
            inst = self._images(self.B)                                            # instance batch (main.py:35)
            inst_label = torch.from_numpy(self.rng.integers(0, self.num_class, size=self.B).astype(np.int64))
            ids = torch.from_numpy(self.rng.choice(self.num_class, size=h, replace=False).astype(np.int64))
            yield inst, inst_label, self._images(h), self._images(h), ids         # + id batch (main.py:43)


def save_checkpoint(path, ffc_net, pool, optimizer=None, real_iter=0, allocator=True):
    """The reference's dictionary (main.py:85: state_dict / lru / fc / qp) plus one extra key, `resume`, with what
    the reference does not save but an exact continuation needs (EMA'd gallery weights, optimizer momenta).
    allocator=False (shard-wise checkpoints): `lru` and `qp` stay None here — every rank's pool file holds them as
    arrays (ShardedFFC.pool_state), and a 10 M-entry list of tuples / dict would cost minutes of host serialisation."""
    torch.save({'state_dict': ffc_net.probe_net.state_dict(), 'lru': ffc_net.lru.state_dict() if allocator else None,
                'fc': pool.cpu() if pool is not None else None,
                'qp': ffc_net.queue_position_dict.to_dict() if allocator else None,
                'resume': {'gallery_state_dict': ffc_net.gallery_net.state_dict(), 'real_iter': int(real_iter),
                           'optimizer': optimizer.state_dict() if optimizer is not None else None}}, path)


def load_checkpoint(path, ffc_net, optimizer=None, step_model=None):
    """Resume (the reference only saves): weights, pool, LRU order via LRU.restore (lru.py:113) and
    queue positions.  A checkpoint written by the reference itself (no `resume` key) restarts the gallery net
    as a copy of the probe net, as FFC.__init__ does (ffc.py:53-55).  With an identity-sharded `step_model` the pool
    comes from this rank's `<name>.pool<rank>.pt` beside `path` (what a sharded run writes), or — for a file that
    holds the whole pool (`fc`) — from this rank's slot range of it.  Returns the iteration to continue from."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    ffc_net.probe_net.load_state_dict(ck['state_dict'])
    extra = ck.get('resume') or {}
    ffc_net.gallery_net.load_state_dict(extra.get('gallery_state_dict') or ck['state_dict'])
    sharded = step_model is not None and hasattr(step_model, 'load_pool_state')
    if ck.get('fc') is None:
        if not sharded:
            raise ValueError("%s holds no pool (written by a sharded run): resume it with the same number of ranks" % path)
        shard_file = '%s.pool%d.pt' % (path[:-3] if path.endswith('.pt') else path, step_model.rank)
        step_model.load_pool_state(torch.load(shard_file, map_location="cpu", weights_only=True))
    else:
        with torch.no_grad():
            if sharded:
                Qs = step_model.head.queue.shape[1]
                step_model.head.queue.copy_(ck['fc'][:, step_model.rank * Qs:(step_model.rank + 1) * Qs])
            else:
                ffc_net.queue.copy_(ck['fc'].to(ffc_net.queue.device))
        state = ffc_net._state()
        state.lru.reset()
        state.lru.restore([tuple(kv) for kv in ck['lru']])
        qp = ck['qp']
        state.qp[:] = np.asarray([qp[i] for i in range(len(qp))], dtype=np.uint8)
    if optimizer is not None and extra.get('optimizer') is not None:
        optimizer.load_state_dict(extra['optimizer'])
        if hasattr(optimizer, 'scatter_state'):
            optimizer.scatter_state()                    # this rank's 1 / world momentum slices ...
            optimizer.release_consolidated()             # ... and the full-size buffers go again
    return int(extra.get('real_iter', 0))


def train_one_epoch(data, ffc_net, step_model, optimizer, cur_epoch, conf, real_iter, lr_policy, lr_scheduler,
                    max_epochs, world=1, log=print, skip=0):
    random.seed(cur_epoch)
    db_size = len(data)
    start = time.time()
    loss = None
    for batch_idx, (ins_images, instance_label, images1, images2, id_indexes) in enumerate(data):
        if batch_idx < skip:                                                       # resumed inside this epoch
            continue
        if lr_policy != 'ReduceLROnPlateau':
            lr_scheduler.update(None, batch_idx * 1.0 / db_size)                   # main.py:39-40
        inst1, inst2 = torch.chunk(ins_images, 2)                                  # main.py:53-54
        lab1, lab2 = torch.chunk(instance_label, 2)
        optimizer.zero_grad()
        x = torch.cat([images1, inst1])                                            # main.py:57-60
        y = torch.cat([images2, inst2])
        x_label = torch.cat([id_indexes, lab1])
        y_label = torch.cat([id_indexes, lab2])
        loss = step_model(x, y, x_label, y_label)                                  # main.py:65
        loss.backward()
        if world > 1:
            step_model.reduce_gradients(optimizer)
        optimizer.step()
        real_iter += 1
        if real_iter % conf.print_freq == 0:                                       # main.py:76-85
            # N > 1: every rank holds its own rows' share of the loss; the reference's scalar is their sum (a collective)
            loss_val = float(step_model.global_loss(loss)) if world > 1 else loss.item()
            lr = optimizer.param_groups[0]['lr']
            log("epoch %d iter %d loss %.4f lr %.5f  %.1f it/s" % (cur_epoch, real_iter, loss_val, lr,
                                                                   conf.print_freq / max(time.time() - start, 1e-9)))
            if lr_policy == 'ReduceLROnPlateau':
                lr_scheduler.step(loss_val)
            start = time.time()
            tag = real_iter // conf.print_freq
            if conf.saved_dir:
                # the partitioned optimizer's momenta live as 1/N slices: gather them into the per-parameter entries
                # torch's state_dict() serialises (a collective: every rank, whichever pool form), and let them go again
                # once the file is written — only when a checkpoint is written at all
                if hasattr(optimizer, 'consolidate_state'):
                    optimizer.consolidate_state()
                os.makedirs(conf.saved_dir, exist_ok=True)
                if hasattr(step_model, 'pool_state'):
                    # sharded pool: every rank writes ITS slots and the (replicated) allocator state as arrays (no rank
                    # ever holds the whole pool: 410 GB at 100 M identities), rank 0 the model file with fc / lru / qp = None
                    torch.save(step_model.pool_state(), os.path.join(conf.saved_dir, '%d.pool%d.pt' % (tag, step_model.rank)))
                    if step_model.rank == 0:
                        save_checkpoint(os.path.join(conf.saved_dir, '%d.pt' % tag), ffc_net, None, optimizer, real_iter,
                                        allocator=False)
                elif world == 1 or torch.distributed.get_rank() == 0:
                    save_checkpoint(os.path.join(conf.saved_dir, '%d.pt' % tag), ffc_net, ffc_net.queue, optimizer, real_iter)
                if hasattr(optimizer, 'release_consolidated'):
                    optimizer.release_consolidated()
    return real_iter, loss


def train(conf, log=print):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            from .parallel import warm_stream_pool
            warm_stream_pool(dev)                                      # before RCCL's streams exist, see its docstring
            backend = getattr(conf, "dist_backend", "nccl")            # "gloo": rehearsal on one GPU (tests), host-staged
            dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    torch.manual_seed(0)
    sharded = world > 1 and conf.queue_size % world == 0 and getattr(conf, 'pool', 'sharded') == 'sharded'
    ffc_net = FFC(conf.net_type, conf.feat_dim, conf.queue_size, conf.scale, conf.loss_type, conf.margin, conf.alpha,
                  conf.neg_margin, conf.pretrained_model_path, conf.num_class,           # main.py:116-117
                  pool_device=dev if world > 1 else None,                                # N > 1: no rank ever builds or
                  pool_shard=(int(os.environ.get("RANK", "0")), world) if sharded else None).cuda()   # holds the whole pool
    if getattr(conf, "head_dtype", "bf16") != "bf16":
        ffc_net.head_dtype = conf.head_dtype                                             # "fp8": the e4m3 class matmul (config C5)
    optim_config = load_config(conf.optim_config) if conf.optim_config else dict(OPTIM_CONFIG)
    step_model = ffc_net
    start_iter = 0
    if world > 1:
        # one process per GPU: identity-sharded pool + partitioned SGD (parallel.py); the scheduler keeps the
        # reference's interface around the partitioned optimizer
        from .parallel import DataParallelFFC, ShardedFFC
        from .optim.optimizer import WarmupSchedule
        step_model = ShardedFFC(ffc_net, dist) if sharded else DataParallelFFC(ffc_net, dist)
        if optim_config['optim'] != 'SGD' or optim_config['scheduler'] == 'ReduceLROnPlateau':
            raise ValueError("multi-GPU runs use the partitioned SGD with a warm-up schedule")
        optim = step_model.make_optimizer(optim_config['LR'], optim_config['momentum'], optim_config['decay'],
                                          optim_config['nesterov'])
        hyper = {k: optim_config[k] for k in ('milestones', 'gammas', 'eta_min', 'gamma') if k in optim_config}
        if optim_config['scheduler'] == 'cos':
            hyper = dict(T_max=optim_config['epochs'], eta_min=optim_config['eta_min'])
        elif optim_config['scheduler'] == 'linear':
            hyper = dict(max_LR=optim_config['LR'], min_LR=optim_config['LR_min'])
        elif optim_config['scheduler'] == 'multistep':
            hyper = dict(milestones=optim_config['milestones'], gammas=optim_config['gammas'])
        elif optim_config['scheduler'] == 'exponential':
            hyper = dict(gamma=optim_config['gamma'])
        lr_scheduler = WarmupSchedule(optim, optim_config['scheduler'], optim_config.get('warmup', 0),
                                      optim_config.get('epochs', 1), **hyper)
        if getattr(conf, "resume", ""):
            start_iter = load_checkpoint(conf.resume, ffc_net, optim, step_model)
    else:
        optim, lr_scheduler = get_optim_scheduler([p for p in ffc_net.parameters() if p.requires_grad], optim_config)
        if getattr(conf, "resume", ""):
            start_iter = load_checkpoint(conf.resume, ffc_net, optim)
    rank = dist.get_rank() if dist else 0
    real_iter, loss = start_iter, None
    for epoch in range(optim_config['epochs']):                                    # main.py:134-140
        if start_iter >= (epoch + 1) * conf.iters_per_epoch:
            continue
        if optim_config['scheduler'] != 'ReduceLROnPlateau':
            lr_scheduler.update(epoch, 0.0)
        if getattr(conf, "data_store", ""):
            # the reference's two datasets (main.py:102-111) over a FaceStore; transform on the GPU (data.py)
            from .data import DeviceBatcher, MultiLMDBDataset, PairLMDBDataset
            stores, kvs = conf.data_store.split(","), conf.data_kv.split(",")
            data = DeviceBatcher(MultiLMDBDataset(stores, kvs), PairLMDBDataset(stores, kvs), conf.batch_size, dev,
                                 n_batches=conf.iters_per_epoch, seed=1000 * epoch + rank)
        else:
            data = SyntheticFaces(conf.num_class, conf.batch_size, conf.iters_per_epoch, dev, seed=1000 * epoch + rank)
        real_iter, loss = train_one_epoch(data, ffc_net, step_model, optim, epoch + 1, conf, real_iter,
                                          optim_config['scheduler'], lr_scheduler, optim_config['epochs'], world, log,
                                          skip=max(0, start_iter - epoch * conf.iters_per_epoch))
    return ffc_net, loss


def parse_args(argv=None):
    conf = argparse.ArgumentParser(description='fast face classification (MI355X path).')
    conf.add_argument('--saved_dir', default='checkpoint', type=str)
    conf.add_argument('--net_type', type=str, default='ir50')            # the reference's default 'r50' is out of scope
    conf.add_argument('--queue_size', type=int, default=1000)
    conf.add_argument('--print_freq', type=int, default=1000)
    conf.add_argument('--pretrained_model_path', type=str, default='')
    conf.add_argument('--batch_size', type=int, default=64)
    conf.add_argument('--alpha', type=float, default=0.99)
    conf.add_argument('--loss_type', type=str, default='Arc', choices=['Arc', 'AM', 'SV'])
    conf.add_argument('--margin', type=float, default=0.5)
    conf.add_argument('--scale', type=float, default=32.0)
    conf.add_argument('--neg_margin', type=float, default=0.25)
    conf.add_argument('--sync_bn', action='store_true', default=False)   # parsed and unused, as in the reference (:162)
    conf.add_argument('--feat_dim', type=int, default=512)
    conf.add_argument('--num_class', type=int, default=100000, help='identities of the synthetic dataset')
    conf.add_argument('--iters_per_epoch', type=int, default=100)
    conf.add_argument('--optim_config', type=str, default='', help='typed-JSON file in the format of config/optim_config')
    conf.add_argument('--data_store', type=str, default='', help='comma-separated FaceStore directories (data.py; the reference '
                      'hard-codes its LMDB paths at main.py:168-169); empty = synthetic batches')
    conf.add_argument('--data_kv', type=str, default='', help='the kv files of --data_store ("<key> <label>" lines)')
    conf.add_argument('--head_dtype', type=str, default='bf16', choices=['bf16', 'fp8'],
                      help='operand type of the class matmul (fp8: e4m3 shadow of the pool, csrc/head8.hip)')
    conf.add_argument('--dist_backend', type=str, default='nccl', choices=['nccl', 'gloo'])
    conf.add_argument('--pool', type=str, default='sharded', choices=['sharded', 'replicated'],
                      help='N > 1: identity-sharded pool (default) or a full replica per rank')
    conf.add_argument('--resume', type=str, default='', help='checkpoint written by this driver (or by the reference) to continue from')
    return conf.parse_args(argv)


if __name__ == '__main__':
    train(parse_args())
