"""Optimizer + learning-rate schedule factory honouring the reference's contract
(optim/optimizer.py:6-45 scheduler interface, :142-168 factory):

    optimizer, scheduler = get_optim_scheduler(parameters, config_dict)
    scheduler.update(cur_epoch: int | None, cur_iter: float | None)      # main.py:39-40,138-139
    scheduler.get_lr() -> list;  scheduler.state_dict() / load_state_dict()

The SGD branch returns the fused gfx950 step (optim/fused.py) behind the torch Optimizer interface.
The four warm-up schedules are one class parameterised by a decay rule (host arithmetic only).
"""
import math
from bisect import bisect_right

import torch
from torch.optim import Optimizer

from .fused import FusedSGD


def _rule_multistep(s, base, e):          # optim/optimizer.py:68-89
    lr = base
    for g in s.gammas[:bisect_right(s.milestones, e)]:
        lr *= g
    return lr


def _rule_cosine(s, base, e):             # optim/optimizer.py:47-66 (the stray print at :64 is not reproduced)
    return s.eta_min + (base - s.eta_min) * (1 + math.cos(math.pi * e / s.T_max)) / 2


def _rule_exponential(s, base, e):        # optim/optimizer.py:91-107
    return base * (s.gamma ** e)


def _rule_linear(s, base, e):             # optim/optimizer.py:109-128
    return base * (1 - (s.max_LR - s.min_LR) * e / s.max_epochs / s.max_LR)


_RULES = dict(multistep=_rule_multistep, cos=_rule_cosine, exponential=_rule_exponential, linear=_rule_linear)


class WarmupSchedule(object):
    """lr(epoch, iter) = base * (epoch + iter) / warmup while epoch < warmup, then rule(base, epoch - warmup)."""

    def __init__(self, optimizer, kind, warmup_epochs, epochs, **hyper):
        if not isinstance(optimizer, Optimizer):
            raise TypeError('{:} is not an Optimizer'.format(type(optimizer).__name__))
        self.optimizer = optimizer
        self.kind = kind
        self.base_lrs = [g.setdefault('initial_lr', g['lr']) for g in optimizer.param_groups]
        self.max_epochs, self.warmup_epochs = epochs, warmup_epochs
        self.current_epoch, self.current_iter = 0, 0
        self.__dict__.update(hyper)

    def get_lr(self):
        e, w = self.current_epoch, self.warmup_epochs
        if e < w:
            return [(e / w + self.current_iter / w) * b for b in self.base_lrs]
        if self.kind == 'cos' and e >= self.max_epochs:
            return [self.eta_min for _ in self.base_lrs]
        return [_RULES[self.kind](self, b, e - w) for b in self.base_lrs]

    def update(self, cur_epoch, cur_iter):
        if cur_epoch is not None:
            assert isinstance(cur_epoch, int) and cur_epoch >= 0, 'invalid cur-epoch : {:}'.format(cur_epoch)
            self.current_epoch = cur_epoch
        if cur_iter is not None:
            assert isinstance(cur_iter, float) and cur_iter >= 0, 'invalid cur-iter : {:}'.format(cur_iter)
            self.current_iter = cur_iter
        for group, lr in zip(self.optimizer.param_groups, self.get_lr()):
            group['lr'] = lr

    def get_min_lr(self):
        return min(self.get_lr())

    def get_min_info(self):
        lrs = self.get_lr()
        return '#LR=[{:.6f}~{:.6f}] epoch={:03d}, iter={:4.2f}#'.format(min(lrs), max(lrs), self.current_epoch,
                                                                        self.current_iter)

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_state_dict(self, state):
        self.__dict__.update(state)

    def __repr__(self):
        return 'WarmupSchedule({}, warmup={}, epochs={}, epoch={}, iter={:.2f})'.format(
            self.kind, self.warmup_epochs, self.max_epochs, self.current_epoch, self.current_iter)


def get_optim_scheduler(parameters, config):
    assert 'optim' in config and 'scheduler' in config, \
        'config must have optim / scheduler / criterion keys instead of {:}'.format(config)
    name = config['optim']
    if name == 'SGD':                       # optim/optimizer.py:148-150
        optim = FusedSGD(parameters, config['LR'], momentum=config['momentum'], weight_decay=config['decay'],
                         nesterov=config['nesterov'])
    elif name == 'RMSprop':                 # :151-152 (not on the hot path: stock torch)
        optim = torch.optim.RMSprop(parameters, config['LR'], momentum=config['momentum'],
                                    weight_decay=config['decay'])
    else:
        raise ValueError('invalid optim : {:}'.format(name))
    kind, warm, epochs = config['scheduler'], config.get('warmup', 0), config.get('epochs', 1)
    if kind == 'multistep':
        assert len(config['milestones']) == len(config['gammas']), 'invalid {:} vs {:}'.format(
            len(config['milestones']), len(config['gammas']))
        sched = WarmupSchedule(optim, kind, warm, epochs, milestones=config['milestones'], gammas=config['gammas'])
    elif kind == 'cos':
        # the reference reads T_max with getattr() on a dict, which always yields `epochs` (:156)
        sched = WarmupSchedule(optim, kind, warm, epochs, T_max=epochs, eta_min=config['eta_min'])
    elif kind == 'exponential':
        sched = WarmupSchedule(optim, kind, warm, epochs, gamma=config['gamma'])
    elif kind == 'linear':
        sched = WarmupSchedule(optim, kind, warm, epochs, max_LR=config['LR'], min_LR=config['LR_min'])
    elif kind == 'ReduceLROnPlateau':
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(optim, patience=config['patience'], min_lr=config['LR_min'])
    else:
        raise ValueError('invalid scheduler : {:}'.format(kind))
    return optim, sched
