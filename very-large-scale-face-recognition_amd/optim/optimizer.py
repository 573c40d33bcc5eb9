"""Optimizer + LR schedule factory with the reference's contract (optim/optimizer.py:6-45,142-168):
``get_optim_scheduler(parameters, config) -> (optimizer, scheduler)``; schedulers expose
``update(cur_epoch, cur_iter)``, ``get_lr()``, ``state_dict()`` / ``load_state_dict()``.
The SGD branch returns the fused gfx950 step (optim/fused.py) behind the torch Optimizer interface;
the schedules are host arithmetic."""
import math
from bisect import bisect_right

import torch
from torch.optim import Optimizer

from .fused import FusedSGD


class WarmupSchedule(object):
    """Base of the four schedules: linear warm-up over `warmup_epochs`, then `_after(base_lr, e)`
    with e = epochs since warm-up (reference _LRScheduler, optim/optimizer.py:6-45)."""

    def __init__(self, optimizer, warmup_epochs, epochs):
        if not isinstance(optimizer, Optimizer):
            raise TypeError('{:} is not an Optimizer'.format(type(optimizer).__name__))
        self.optimizer = optimizer
        for group in optimizer.param_groups:
            group.setdefault('initial_lr', group['lr'])
        self.base_lrs = [group['initial_lr'] for group in optimizer.param_groups]
        self.max_epochs = epochs
        self.warmup_epochs = warmup_epochs
        self.current_epoch = 0
        self.current_iter = 0

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)

    def _after(self, base_lr, e):
        raise NotImplementedError

    def get_lr(self):
        if self.current_epoch >= self.warmup_epochs:
            return [self._after(b, self.current_epoch - self.warmup_epochs) for b in self.base_lrs]
        frac = self.current_epoch / self.warmup_epochs + self.current_iter / self.warmup_epochs
        return [frac * b for b in self.base_lrs]

    def get_min_lr(self):
        return min(self.get_lr())

    def get_min_info(self):
        lrs = self.get_lr()
        return '#LR=[{:.6f}~{:.6f}] epoch={:03d}, iter={:4.2f}#'.format(min(lrs), max(lrs), self.current_epoch,
                                                                        self.current_iter)

    def __repr__(self):
        return '{:}(warmup={:}, max-epoch={:}, current::epoch={:}, iter={:.2f})'.format(
            self.__class__.__name__, self.warmup_epochs, self.max_epochs, self.current_epoch, self.current_iter)

    def update(self, cur_epoch, cur_iter):
        if cur_epoch is not None:
            assert isinstance(cur_epoch, int) and cur_epoch >= 0, 'invalid cur-epoch : {:}'.format(cur_epoch)
            self.current_epoch = cur_epoch
        if cur_iter is not None:
            assert isinstance(cur_iter, float) and cur_iter >= 0, 'invalid cur-iter : {:}'.format(cur_iter)
            self.current_iter = cur_iter
        for group, lr in zip(self.optimizer.param_groups, self.get_lr()):
            group['lr'] = lr


class CosineAnnealingLR(WarmupSchedule):       # optim/optimizer.py:47-66 (without the stray print at :64)
    def __init__(self, optimizer, warmup_epochs, epochs, T_max, eta_min):
        self.T_max, self.eta_min = T_max, eta_min
        super(CosineAnnealingLR, self).__init__(optimizer, warmup_epochs, epochs)

    def get_lr(self):
        if self.current_epoch >= self.max_epochs and self.current_epoch >= self.warmup_epochs:
            return [self.eta_min for _ in self.base_lrs]
        return super(CosineAnnealingLR, self).get_lr()

    def _after(self, base_lr, e):
        return self.eta_min + (base_lr - self.eta_min) * (1 + math.cos(math.pi * e / self.T_max)) / 2


class MultiStepLR(WarmupSchedule):             # optim/optimizer.py:68-89
    def __init__(self, optimizer, warmup_epochs, epochs, milestones, gammas):
        assert len(milestones) == len(gammas), 'invalid {:} vs {:}'.format(len(milestones), len(gammas))
        self.milestones, self.gammas = milestones, gammas
        super(MultiStepLR, self).__init__(optimizer, warmup_epochs, epochs)

    def _after(self, base_lr, e):
        lr = base_lr
        for g in self.gammas[:bisect_right(self.milestones, e)]:
            lr *= g
        return lr


class ExponentialLR(WarmupSchedule):           # optim/optimizer.py:91-107
    def __init__(self, optimizer, warmup_epochs, epochs, gamma):
        self.gamma = gamma
        super(ExponentialLR, self).__init__(optimizer, warmup_epochs, epochs)

    def _after(self, base_lr, e):
        return base_lr * (self.gamma ** e)


class LinearLR(WarmupSchedule):                # optim/optimizer.py:109-128
    def __init__(self, optimizer, warmup_epochs, epochs, max_LR, min_LR):
        self.max_LR, self.min_LR = max_LR, min_LR
        super(LinearLR, self).__init__(optimizer, warmup_epochs, epochs)

    def _after(self, base_lr, e):
        return base_lr * (1 - (self.max_LR - self.min_LR) * e / self.max_epochs / self.max_LR)


def get_optim_scheduler(parameters, config):   # optim/optimizer.py:142-168
    assert 'optim' in config and 'scheduler' in config, \
        'config must have optim / scheduler / criterion keys instead of {:}'.format(config)
    if config['optim'] == 'SGD':
        optim = FusedSGD(parameters, config['LR'], momentum=config['momentum'], weight_decay=config['decay'],
                         nesterov=config['nesterov'])
    elif config['optim'] == 'RMSprop':
        optim = torch.optim.RMSprop(parameters, config['LR'], momentum=config['momentum'],
                                    weight_decay=config['decay'])
    else:
        raise ValueError('invalid optim : {:}'.format(config['optim']))
    sched = config['scheduler']
    if sched == 'cos':
        T_max = getattr(config, 'T_max', config['epochs'])   # a dict has no such attribute: always `epochs` (:156)
        scheduler = CosineAnnealingLR(optim, config['warmup'], config['epochs'], T_max, config['eta_min'])
    elif sched == 'multistep':
        scheduler = MultiStepLR(optim, config['warmup'], config['epochs'], config['milestones'], config['gammas'])
    elif sched == 'exponential':
        scheduler = ExponentialLR(optim, config['warmup'], config['epochs'], config['gamma'])
    elif sched == 'linear':
        scheduler = LinearLR(optim, config['warmup'], config['epochs'], config['LR'], config['LR_min'])
    elif sched == 'ReduceLROnPlateau':
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optim, patience=config['patience'],
                                                               min_lr=config['LR_min'])
    else:
        raise ValueError('invalid scheduler : {:}'.format(sched))
    return optim, scheduler
