"""Learning-rate schedule used by the reference's train loop (gluoncv.utils.LRScheduler / LRSequential,
constructed at train_yolov3.py:517-525, evaluated once per iteration by the Trainer).

[UPSTREAM-UNVERIFIED] (GluonCV is not vendored): semantics per SURVEY.md A.4 —
  'linear'   lr = base + (target - base) * T / N          (warm-up; N = nepochs * iters_per_epoch)
  'constant' lr = base
  'step'     lr = base * step_factor ** (#step_iter <= T)  with step_iter = step_epoch * iters_per_epoch
  'poly'     lr = target + (base - target) * (1 - T/N) ** power
  'cosine'   lr = target + (base - target) * (1 + cos(pi * T/N)) / 2
LRSequential chains schedulers, each consuming its own nepochs * iters_per_epoch updates.
"""
import math


class LRScheduler:
    def __init__(self, mode, base_lr=0.1, target_lr=0, niters=0, nepochs=0, iters_per_epoch=0, offset=0, power=2,
                 step_iter=None, step_epoch=None, step_factor=0.1):
        assert mode in ("constant", "step", "linear", "poly", "cosine")
        self.mode, self.base_lr = mode, base_lr
        self.target_lr = base_lr if mode == "constant" else target_lr
        self.niters = niters if niters > 0 else nepochs * iters_per_epoch
        if mode == "step":
            if step_iter is None:
                step_iter = [int(s) * iters_per_epoch for s in (step_epoch or [])]
            self.step = list(step_iter)
        self.step_factor, self.power, self.offset = step_factor, power, offset
        self.learning_rate = base_lr

    def __call__(self, num_update):
        self.update(num_update)
        return self.learning_rate

    def update(self, num_update):
        N = max(self.niters - 1, 1)
        T = min(max(0, num_update - self.offset), N)
        if self.mode == "constant":
            factor = 0.0
        elif self.mode == "linear":
            factor = 1 - T / N
        elif self.mode == "poly":
            factor = (1 - T / N) ** self.power
        elif self.mode == "cosine":
            factor = (1 + math.cos(math.pi * T / N)) / 2
        if self.mode == "step":
            count = sum(1 for s in self.step if s <= T)
            self.learning_rate = self.base_lr * (self.step_factor ** count)
        else:
            self.learning_rate = self.target_lr + (self.base_lr - self.target_lr) * factor


class LRSequential:
    def __init__(self, schedulers):
        self.schedulers, self.update_sep, self.count = [], [], 0
        for s in schedulers:
            if s.niters <= 0:
                continue                       # e.g. zero warm-up epochs
            s.offset = self.count
            self.count += s.niters
            self.update_sep.append(self.count)
            self.schedulers.append(s)
        self.learning_rate = self.schedulers[0].base_lr if self.schedulers else 0.0

    def __call__(self, num_update):
        num_update = min(num_update, self.count - 1)
        for s, sep in zip(self.schedulers, self.update_sep):
            if num_update < sep:
                self.learning_rate = s(num_update)
                return self.learning_rate
        self.learning_rate = self.schedulers[-1](num_update)
        return self.learning_rate
