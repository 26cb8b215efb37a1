"""``gluon.Trainer`` for the one optimizer the reference uses on this path: SGD with momentum and
weight decay (train_yolov3.py:527-530 ``Trainer(net.collect_params(), 'sgd', {'wd', 'momentum',
'lr_scheduler'}, kvstore='local')``, ``trainer.step(batch_size)`` :634).

Data parallelism is one process per GPU: ``step`` all-reduces (sum) the flat gradient buffer over
RCCL (``torch.distributed``, backend "nccl") when a process group is initialised — the MI355X
replacement for kvstore='local' reduce + broadcast — then every rank applies the same
update, with ``rescale_grad = 1/batch_size`` (batch_size = the GLOBAL batch, as in the reference).
"""
from . import parallel


class Trainer(object):
    def __init__(self, params, optimizer='sgd', optimizer_params=None, kvstore='device',
                 compression_params=None, update_on_kvstore=None):
        if optimizer != 'sgd':
            raise NotImplementedError("only 'sgd' is on the reference's path (train_yolov3.py:527)")
        self._net = params._net
        op = dict(optimizer_params or {})
        self._lr = float(op.pop('learning_rate', 0.01))
        self._momentum = float(op.pop('momentum', 0.0))
        self._wd = float(op.pop('wd', 0.0))
        self._sched = op.pop('lr_scheduler', None)
        if op:
            raise ValueError("unsupported optimizer_params: %s" % sorted(op))
        self._num_update = 0
        self._kvstore = kvstore
        self._overlap = None

    @property
    def learning_rate(self):
        if self._sched is not None:
            return float(self._sched(self._num_update))
        return self._lr

    def set_learning_rate(self, lr):
        if self._sched is not None:
            raise UserWarning("LRScheduler of the optimizer has already been defined")
        self._lr = float(lr)

    def enable_overlap(self):
        """Overlap the gradient all-reduce with the backward pass (bucketed, side stream)."""
        self._overlap = parallel.GradBucketOverlap(self._net)

    def allreduce_grads(self):
        if self._overlap is not None:
            self._overlap.finish()
        else:
            parallel.allreduce_(self._net._grads)

    def step(self, batch_size, ignore_stale_grad=False):
        net = self._net
        if net._grads is None:
            raise RuntimeError("step() before any recorded forward/backward")
        self._num_update += 1
        lr = self.learning_rate
        net._sync_opts()
        self.allreduce_grads()
        net.sgd_step(lr, self._momentum, self._wd, 1.0 / float(batch_size))
