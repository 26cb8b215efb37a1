"""Mode switches with the semantics of ``mxnet.autograd`` that the reference's model object keys
its behaviour on (models/definitions/yolo/yolo3.py:179, 1143, 1179-1192: ``autograd.is_training()``
/ ``autograd.is_recording()``; callers: train_yolov3.py:623 ``autograd.record()``,
models/definitions/yolo/transforms.py:192 ``autograd.train_mode()``)."""
import contextlib
import threading

_state = threading.local()


def _get():
    if not hasattr(_state, "recording"):
        _state.recording = False
        _state.training = False
    return _state


def is_recording():
    return _get().recording


def is_training():
    return _get().training


@contextlib.contextmanager
def _scope(recording, training):
    s = _get()
    prev = (s.recording, s.training)
    # only record() scopes count: a record() nested in train_mode() / pause() is still an outermost recording
    rec_depth = getattr(s, "rec_depth", 0)
    if recording:
        if rec_depth == 0:
            s.tape = []  # an outermost record(): forwards recorded earlier and never back-propagated are dropped
        s.rec_depth = rec_depth + 1
    if recording is not None:
        s.recording = recording
    if training is not None:
        s.training = training
    try:
        yield
    finally:
        s.recording, s.training = prev
        s.rec_depth = rec_depth


def record(train_mode=True):
    """``with autograd.record():`` — recording on, training mode on (mxnet default)."""
    return _scope(True, train_mode)


def pause(train_mode=False):
    return _scope(False, train_mode)


def train_mode():
    return _scope(None, True)


def predict_mode():
    return _scope(None, False)


# ---- the tape: nets that ran a recorded forward and still owe a backward pass
def _register(net):
    s = _get()
    if not hasattr(s, "tape"):
        s.tape = []
    if net not in s.tape:
        s.tape.append(net)


_loss_cls = None


def loss_vector(tensor, net, index):
    """Wrap one of a net's four (B,) loss vectors so that ``backward`` can tell what its heads are made
    of: a torch.Tensor subclass whose ``terms`` ({(id(net), loss index): multiplicity}) survive ``+``
    (``obj + center + scale + cls``, ``sum([...])``) and nothing else — any other operation returns a
    plain tensor."""
    global _loss_cls
    if _loss_cls is None:
        import collections
        import torch

        class LossVector(torch.Tensor):
            terms = None

            @classmethod
            def __torch_function__(cls, func, types, args=(), kwargs=None):
                with torch._C.DisableTorchFunctionSubclass():
                    out = func(*args, **(kwargs or {}))
                if not isinstance(out, torch.Tensor):
                    return out
                out = out.as_subclass(torch.Tensor)
                if func in (torch.add, torch.Tensor.add, torch.Tensor.__add__, torch.Tensor.__radd__) and not kwargs:
                    terms = collections.Counter()
                    for a in args:
                        if isinstance(a, LossVector) and a.terms is not None:
                            terms.update(a.terms)
                        elif not (isinstance(a, (int, float)) and a == 0):
                            return out
                    out = out.as_subclass(LossVector)
                    out.terms = terms
                return out
        _loss_cls = LossVector
    import collections
    out = tensor.as_subclass(_loss_cls)
    out.terms = collections.Counter({(id(net), index): 1})
    return out


def backward(heads, head_grads=None, retain_graph=False, train_mode=True):
    """``autograd.backward(sum_losses)`` (train_yolov3.py:626-631).  The one pattern the reference uses —
    unit head gradients on ``obj_loss + center_loss + scale_loss + cls_loss`` of each net — is what the
    recorded forward already prepared (d(sum of the four)/d(raw predictions)); this checks that `heads` IS
    that sum for every pending net and walks their backward passes.  Anything else (a subset of the
    losses, scaled losses, head_grads) raises instead of silently back-propagating the full sum."""
    if head_grads is not None:
        raise NotImplementedError("non-unit head gradients are not part of the reference's call pattern")
    s = _get()
    tape = getattr(s, "tape", [])
    if not tape:
        raise RuntimeError("autograd.backward() without a recorded forward")
    import collections
    got = collections.Counter()
    for h in (heads if isinstance(heads, (list, tuple)) else [heads]):
        terms = getattr(h, "terms", None)
        if terms is None:
            raise NotImplementedError(
                "autograd.backward: a head is not a sum of a net's loss vectors (built with anything but '+'); "
                "only backward(obj_loss + center_loss + scale_loss + cls_loss) is supported")
        got.update(terms)
    nets = [net for net in tape if any(nid == id(net) for nid, _ in got)]
    want = collections.Counter({(id(net), i): 1 for net in nets for i in range(4)})
    if not nets or got != want:
        raise NotImplementedError(
            "autograd.backward: heads must be exactly obj_loss + center_loss + scale_loss + cls_loss of a recorded "
            "net (train_yolov3.py:626); got loss indices %s" % sorted(i for (_, i), _ in got.items()))
    s.tape = [net for net in tape if net not in nets]
    for net in nets:
        net.backward()
