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
        if compression_params is not None:
            raise NotImplementedError("gradient compression (compression_params) is not on the reference's path "
                                      "(train_yolov3.py:527-530 passes none)")
        if update_on_kvstore:
            raise NotImplementedError("update_on_kvstore=True: every rank applies the update itself after the "
                                      "all-reduce (the reference's kvstore='local' resolves to the same)")
        # kvstore init: every device starts from the same parameters (rank 0's); if the parameters are not
        # on the device yet the recorded forward does it (model.forward_train).  Every rank constructs its
        # Trainer, so this is also where the host-side (gloo) control group is created.
        parallel.make_host_group()
        parallel.sync_replicas(self._net)

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
        if self._overlap is None:
            self._overlap = parallel.GradBucketOverlap(self._net)

    def disable_overlap(self):
        if self._overlap is not None:
            self._overlap.remove()
            self._overlap = None

    def allreduce_grads(self):
        """gluon.Trainer.allreduce_grads: sum the gradients over all ranks (after backward, before update)."""
        if self._net._grads is None:
            raise RuntimeError("allreduce_grads() before any recorded forward/backward")
        if self._overlap is not None:
            self._overlap.finish()
        else:
            parallel.allreduce_(self._net._grads)

    def update(self, batch_size, ignore_stale_grad=False):
        """gluon.Trainer.update: the optimizer step on already all-reduced gradients."""
        net = self._net
        if net._grads is None:
            raise RuntimeError("update() before any recorded forward/backward")
        if float(batch_size) <= 0:
            raise ValueError("batch_size must be positive")
        self._num_update += 1
        lr = self.learning_rate
        net._sync_opts()  # no-op unless a Parameter's lr_mult / wd_mult / grad_req changed
        net.sgd_step(lr, self._momentum, self._wd, 1.0 / float(batch_size))

    def step(self, batch_size, ignore_stale_grad=False):
        """trainer.step(batch_size) (train_yolov3.py:634) = allreduce_grads() + update(batch_size);
        batch_size is the GLOBAL batch (rescale_grad = 1/batch_size)."""
        self.allreduce_grads()
        self.update(batch_size, ignore_stale_grad)
