"""Learning-rate schedules the reference hands to its Trainer (train_yolov3.py:517-530):

    lr_scheduler = LRSequential([
        LRScheduler('linear', base_lr=0, target_lr=lr, nepochs=warmup_epochs, iters_per_epoch=n),
        LRScheduler(lr_mode, base_lr=lr, nepochs=epochs - warmup_epochs - start_epoch, iters_per_epoch=n,
                    step_epoch=lr_decay_epoch, step_factor=lr_decay, power=2)])
    Trainer(params, 'sgd', {'wd': .., 'momentum': .., 'lr_scheduler': lr_scheduler})

Both classes come from gluoncv.utils (not in the reference tree, not installable here): their semantics
are restated from the published implementation [UPSTREAM-RECALLED, UNVERIFIED against a gluoncv install]
and pinned by hand-derived values in tests/test_model_host.py.  A scheduler is called with the number of
updates so far and returns the learning rate; ``videoyolo_amd.Trainer`` calls it once per ``step``.
"""
import math


class LRScheduler(object):
    """mode: 'constant' | 'step' | 'linear' | 'poly' | 'cosine'.
    For every mode but 'step':  lr = target_lr + (base_lr - target_lr) * factor(T / (niters - 1)),
    with T = clip(num_update - offset, 0, niters - 1) and factor = 1 (constant), 1 - t (linear),
    (1 - t)^power (poly), (1 + cos(pi t)) / 2 (cosine).  'step': lr = base_lr * step_factor^k, k = number of
    step boundaries (in iterations) <= T."""

    def __init__(self, mode, base_lr=0.1, target_lr=0, niters=0, nepochs=0, iters_per_epoch=0, offset=0,
                 power=2, step_iter=None, step_epoch=None, step_factor=0.1):
        if mode not in ('constant', 'step', 'linear', 'poly', 'cosine'):
            raise ValueError("unknown lr mode %r" % (mode,))
        self.mode = mode
        self.base_lr = base_lr
        self.target_lr = base_lr if mode == 'constant' else target_lr
        self.niters = niters
        self.step = step_iter
        epoch_iters = nepochs * iters_per_epoch
        if epoch_iters > 0:
            self.niters = epoch_iters
            if step_epoch is not None:
                self.step = [s * iters_per_epoch for s in step_epoch]
        if mode == 'step' and self.step is None:
            raise ValueError("mode 'step' needs step_iter or step_epoch")
        self.offset = offset
        self.power = power
        self.step_factor = step_factor
        self.learning_rate = base_lr

    def __call__(self, num_update):
        self.update(num_update)
        return self.learning_rate

    def update(self, num_update):
        n = max(self.niters - 1, 0)
        t = min(max(0, num_update - self.offset), n)
        frac = (t / float(n)) if n > 0 else 1.0
        if self.mode == 'constant':
            factor = 0.0
        elif self.mode == 'linear':
            factor = 1.0 - frac
        elif self.mode == 'poly':
            factor = (1.0 - frac) ** self.power
        elif self.mode == 'cosine':
            factor = (1.0 + math.cos(math.pi * frac)) / 2.0
        else:  # step
            count = sum(1 for s in self.step if s <= t)
            self.learning_rate = self.base_lr * (self.step_factor ** count)
            return
        self.learning_rate = self.target_lr + (self.base_lr - self.target_lr) * factor


class LRSequential(object):
    """Runs its schedulers back to back: scheduler i covers the updates [offset_i, offset_i + niters_i)."""

    def __init__(self, schedulers):
        if not schedulers:
            raise ValueError("LRSequential needs at least one scheduler")
        self.update_sep = []
        self.count = 0
        self.learning_rate = 0
        self.schedulers = []
        for s in schedulers:
            s.offset = self.count
            self.count += s.niters
            self.update_sep.append(self.count)
            self.schedulers.append(s)

    def __call__(self, num_update):
        self.update(num_update)
        return self.learning_rate

    def update(self, num_update):
        num_update = min(num_update, self.count - 1)
        ind = len(self.schedulers) - 1
        for i, sep in enumerate(self.update_sep):
            if sep > num_update:
                ind = i
                break
        s = self.schedulers[ind]
        s.update(num_update)
        self.learning_rate = s.learning_rate
