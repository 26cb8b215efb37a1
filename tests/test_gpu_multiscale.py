"""-m gpu: the reference's DEFAULT training mode — the input side changes every `interval` batches
(train_yolov3.py:258-271: `RandomTransformDataLoader([... x * 32 for x in range(10, 20)], interval=10)`).

One net and one trainer walk through a sequence of sizes (the workspace is re-planned at every change, it grows
and it is reused when the size shrinks again).  Two properties, both BIT-exact:

  * the sequence is reproducible: the same sequence from the same parameters gives the same losses, gradients and
    updated parameters, bit for bit (no float atomics anywhere on the step, whatever plan was bound before);
  * a step at size S inside the sequence equals the same step on a FRESH net (fresh workspace, first plan ever) that
    was given the sequence's current parameters: nothing of an earlier size's plan — planes, borders, BatchNorm
    partial-sum slabs, stream-K hand-off slabs, wgrad slabs — leaks into the next size.

Sizes are small multiples of 32 (the full 320 ... 608 sequence at batch 16 is bench.py's also_train_multiscale and
tools/multiscale_check.py); one case uses the real 320 -> 608 -> 416 jump at batch 1 for the large-plane code paths.
"""
import numpy as np
import pytest

from conftest import frames

pytestmark = pytest.mark.gpu

C = 20


def _data(B, S, seed):
    from oracle import targets_oracle as T
    x = frames(B, S, seed=seed)
    gt_boxes, gt_ids = T.synthetic_gt(B, S, C, m=3, seed=seed + 1, pad_to=4)
    return [x, gt_boxes] + list(T.prefetch_targets(C, S, S, gt_boxes, gt_ids))


def _net(params):
    import videoyolo_amd as vy
    net = vy.yolo3_darknet53(["c%d" % i for i in range(C)], pretrained_base=False)
    net.set_parameters(params)
    net.collect_params().reset_ctx("cuda:0")
    return net


def _all_params(net):
    return {p.name: np.array(p.data(), copy=True) for p in net.collect_params().values()}


def _step(net, trainer, data, B):
    import torch
    from videoyolo_amd import autograd
    with autograd.record():
        losses = net(*data)
        autograd.backward([losses[0] + losses[1] + losses[2] + losses[3]])
    torch.cuda.synchronize()
    ls = np.stack([l.cpu().numpy() for l in losses])
    grads = net._grads.clone()
    if trainer is not None:
        trainer.step(B)
    return ls, grads


def _run_sequence(params, seq, B, with_fresh):
    import torch
    import videoyolo_amd as vy
    net = _net(params)
    trainer = vy.Trainer(net.collect_params(), 'sgd', {'learning_rate': 1e-3, 'wd': 5e-4, 'momentum': 0.9})
    out = []
    for i, S in enumerate(seq):
        data = _data(B, S, seed=40 + i)
        if with_fresh:
            fresh = _net(_all_params(net))       # running statistics included
            f_ls, f_gr = _step(fresh, None, data, B)
            del fresh
        ls, gr = _step(net, trainer, data, B)
        if with_fresh:
            assert np.array_equal(ls, f_ls), "step %d (size %d): losses differ from a fresh net's" % (i, S)
            assert torch.equal(gr, f_gr), "step %d (size %d): gradients differ from a fresh net's" % (i, S)
        assert np.isfinite(ls).all()
        out.append((ls, gr.cpu().numpy()))
    final = _all_params(net)
    return out, final


@pytest.mark.parametrize("seq,B", [((96, 160, 64, 128, 96, 160, 160, 64), 2), ((320, 608, 416), 1)])
def test_size_sequence_is_bit_reproducible_and_matches_fresh_nets(seq, B, synth20):
    a, fa = _run_sequence(synth20, seq, B, with_fresh=True)
    b, fb = _run_sequence(synth20, seq, B, with_fresh=False)
    for i, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        assert np.array_equal(la, lb), "step %d: losses of two runs of the sequence differ" % i
        assert np.array_equal(ga, gb), "step %d: gradients of two runs of the sequence differ" % i
    for k in fa:
        assert np.array_equal(fa[k], fb[k]), "parameter %s after the sequence differs between two runs" % k


def test_inference_after_a_training_size_change(synth20):
    """detect -> train at another size -> detect again at the first size: the inference plan is rebuilt over the
    training workspace and returns the rows it returned before (parameters untouched: no trainer step)."""
    from videoyolo_amd import autograd
    net = _net(synth20)
    x = frames(2, 96, seed=3)
    before = [t.cpu().numpy() for t in net(x)]
    data = _data(2, 128, seed=8)
    with autograd.record():
        losses = net(*data)
        autograd.backward([losses[0] + losses[1] + losses[2] + losses[3]])
    # the recorded forward wrote new running statistics: put the old ones back, as a validation pass on the
    # reference's `net` would see the trained ones — here the point is the plan, not the statistics
    net.set_parameters(synth20)
    after = [t.cpu().numpy() for t in net(x)]
    for u, v in zip(before, after):
        assert np.array_equal(u, v)
