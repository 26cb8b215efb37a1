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
    if recording is not None:
        s.recording = recording
    if training is not None:
        s.training = training
    try:
        yield
    finally:
        s.recording, s.training = prev


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


def backward(heads, head_grads=None, retain_graph=False, train_mode=True):
    """``autograd.backward(sum_losses)`` (train_yolov3.py:631).  The only pattern the reference uses —
    unit head gradients on the sum of the four loss vectors of each net — is what the recorded
    forward already prepared; this walks every pending net's backward pass."""
    if head_grads is not None:
        raise NotImplementedError("non-unit head gradients are not part of the reference's call pattern")
    s = _get()
    tape, s.tape = getattr(s, "tape", []), []
    if not tape:
        raise RuntimeError("autograd.backward() without a recorded forward")
    for net in tape:
        net.backward()
