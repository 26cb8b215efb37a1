#!/usr/bin/env python3
"""Captures golden vectors of the hot path FROM THE REFERENCE ITSELF (mxnet + gluoncv + /root/reference) —
the fixtures that turn SURVEY.md §8(c) from "parity unpinned" into "pinned".

This script has NEVER BEEN RUN: mxnet and gluoncv are not installable in the build container (no network, no
wheels), so `tests/golden/mxnet_*.npz` do not exist yet and `tests/test_mxnet_goldens.py` skips.
Anyone with a machine where `import mxnet, gluoncv` works closes the gap with

    python tests/golden/make_mxnet_goldens.py --ref /path/to/VideoYOLO     # writes tests/golden/mxnet_*.npz
    python -m pytest tests/test_mxnet_goldens.py -q                        # oracle vs the goldens (CPU)
    python -m pytest tests/test_mxnet_goldens.py -q -m gpu                 # HIP path vs the goldens (MI355X)

and commits the .npz files (the script and the reference never travel to the GPU box; the fixtures do).

What is captured (all through the reference's own objects — `models/definitions/yolo/wrappers.py:9`
`yolo3_darknet53(classes, pretrained_base=False)`, i.e. `YOLOV3T` at k=1, `yolo3.py:915-1302`):

  mxnet_infer_<S>.npz   S in {416 (BASELINE configs[0]: batch 1, seed 233), 608}: the three raw prediction-conv
                        outputs (forward hooks on `yolo_outputs[i].prediction`, yolo3.py:62), the pre-NMS detection
                        tensor (set_nms(nms_thresh=-1): yolo3.py:1195-1206 returns it un-suppressed), the full-length
                        `box_nms` output (post_nms=-1; yolo3.py:1197-1200) and the (ids, scores, bboxes) of the
                        default call (0.45 / 400 / 100)
  mxnet_infer_416_sparse.npz  the same capture with the objectness biases lowered by 7 (init.synthetic_params obj_bias=-7): a
                        trained-like sparse candidate set (about 50 valid candidates) whose score gaps are mostly wide — where demanding EXACT kept
                        rows is fair (with obj_bias 0 nearly all 212 940 candidates are valid and neighbouring scores of the top
                        400 lie 1e-7 apart: any fp32 summation order other than mxnet's own swaps some of them)
                        Every inference fixture also holds `top/index`, `top/scores`: mxnet's own pre-NMS top 1024 candidates
                        per image in box_nms order, and `nms/first_rows_index`: the pre-NMS row each survivor came from —
                        what tests/test_mxnet_goldens.py needs to tell a near-tie swap (score gap <= 1e-5) from an error
  mxnet_ops.npz         the OPERATOR-LEVEL kit (tests/golden/mxnet_ops_kit.py): one tiny case per recalled semantic choice —
                        box_nms on hand-built rows (thresholds met exactly, duplicate scores in both orders, a top-k cut
                        through a tie, id -1 rows, force_suppress both ways), the same through hand-built LOGITS (the form
                        the HIP path runs: vy_net_detect_heads), box_iou / BBoxBatchIOU on degenerate boxes, the reference's
                        dynamic-target / merger / prefetch generators, YOLOV3Loss on a positive + ignored + negative anchor,
                        one recorded step of the reference's `_conv2d` cell (running_var: biased or not), two SGD steps,
                        imresize(interp=9) at a shrink, an enlarge and a mixed ratio.  tests/test_mxnet_ops.py replays it
  mxnet_train_96.npz    one recorded step on 2 frames of 96x96, 20 classes (train_yolov3.py:623-634): the four (B,)
                        losses (`YOLOV3Loss`, yolo3.py:994,1187), every parameter gradient, the BatchNorm running
                        statistics after the forward, and the parameters after `trainer.step(2)` with
                        SGD(lr 1e-3, momentum 0.9, wd 5e-4)
  mxnet_tiny.params     `save_parameters` of a two-layer gluon net (a few KB): a real mxnet NDArray-dict file for
                        videoyolo_amd/mxparams.py; the full 246 MB yolo3 file is written to --scratch and checked
                        against the reader here, not committed

Parameters are this repo's synthetic ones (videoyolo_amd/init.py `synthetic_params`, seed 233), assigned by
structural name; inputs come from `np.random.RandomState` (a frozen stream).  Nothing big is stored twice: the
fixture holds CRC32s of every input / parameter tensor so that the test can prove it regenerated the same bits.
Large outputs are stored as a seeded sample of elements plus float64 sums.

`--from-oracle DIR` writes fixtures of the SAME layout from this repo's CPU oracle instead — source tag
"oracle-selfcheck", only to exercise the test plumbing (VY_MXNET_GOLDEN_DIR=DIR); such files pin nothing and must
never be committed under tests/golden/.
"""
import argparse
import importlib.util
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
VOC = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
       "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]
SAMPLE = 32768


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


KIT = _load_by_path("mxnet_ops_kit", os.path.join(HERE, "mxnet_ops_kit.py"))
TOP = 1024   # pre-NMS candidates kept per image, in box_nms order (score descending, row ascending)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def frames(b, s, seed):
    """N(0,1) frames from the legacy (frozen) stream: identical bits on every numpy version."""
    return np.random.RandomState(seed).standard_normal((b, 3, s, s)).astype(np.float32)


def sample_idx(size, seed, n=SAMPLE):
    if size <= n:
        return np.arange(size, dtype=np.int64)
    return np.sort(np.random.RandomState(seed).randint(0, size, n).astype(np.int64))


def pack_sampled(out, key, arr, seed, n=SAMPLE):
    """key/shape, key/idx_seed, key/n, key/values (flat sample), key/sum, key/abssum, key/absmax"""
    a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
    out[key + "/shape"] = np.array(arr.shape, np.int64)
    out[key + "/idx_seed"] = np.array(seed, np.int64)
    out[key + "/n"] = np.array(n, np.int64)
    out[key + "/values"] = a[sample_idx(a.size, seed, n)]
    fin = a[np.isfinite(a)].astype(np.float64)
    out[key + "/sum"] = np.array(fin.sum())
    out[key + "/abssum"] = np.array(np.abs(fin).sum())
    out[key + "/absmax"] = np.array(np.abs(fin).max() if fin.size else 0.0)


def synthetic_gt(b, s, c, m, seed):
    rs = np.random.RandomState(seed)
    boxes = np.full((b, m + 1, 4), -1.0, np.float32)       # one padding row (-1), like the batchify pad
    ids = np.full((b, m + 1, 1), -1.0, np.float32)
    for i in range(b):
        for j in range(m):
            w, h = rs.uniform(16, s * 0.6, 2)
            cx, cy = rs.uniform(w / 2, s - w / 2), rs.uniform(h / 2, s - h / 2)
            boxes[i, j] = [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2]
            ids[i, j, 0] = rs.randint(0, c)
    return boxes, ids


# ------------------------------------------------------------------------------------------ backends
class MxnetBackend(object):
    source = "mxnet"

    def __init__(self, ref):
        sys.path.insert(0, ref)
        import mxnet as mx                                                   # noqa: F401
        import gluoncv                                                       # noqa: F401
        from models.definitions.yolo.wrappers import yolo3_darknet53         # wrappers.py:9
        self.mx, self.make = mx, yolo3_darknet53
        self.versions = "mxnet %s, gluoncv %s, numpy %s" % (mx.__version__, gluoncv.__version__, np.__version__)
        self.init = _load_by_path("vy_init", os.path.join(ROOT, "videoyolo_amd", "init.py"))
        self.ops = KIT.MxnetOps(mx, yolo3_darknet53)

    def build(self, classes, size, obj_bias=0.0):
        mx = self.mx
        net = self.make(classes, pretrained_base=False, k=1)
        net.initialize()
        net(mx.nd.zeros((1, 3, size, size)))                                 # deferred shapes -> concrete
        struct = net._collect_params_with_prefix()                           # structural names (save_parameters' keys)
        table = [(k, tuple(v.shape)) for k, v in struct.items()
                 if k.rsplit(".", 1)[1] in ("weight", "bias", "gamma", "beta", "running_mean", "running_var")]
        params = self.init.synthetic_params(table, seed=233, obj_bias=obj_bias)
        for k, v in params.items():
            struct[k].set_data(mx.nd.array(v))
        return net, struct, table, params

    def infer(self, classes, x, obj_bias=0.0):
        mx = self.mx
        net, struct, table, params = self.build(classes, x.shape[2], obj_bias)
        heads = {}
        for i in range(3):                                                   # stride 32, 16, 8 (yolo3.py:1013-1014)
            net.yolo_outputs[i].prediction.register_forward_hook(
                lambda blk, inp, out, i=i: heads.__setitem__(i, out.asnumpy()))
        xin = mx.nd.array(x)
        net.set_nms(nms_thresh=0.45, nms_topk=400, post_nms=100)
        ids, scores, bboxes = [t.asnumpy() for t in net(xin)]
        net.set_nms(nms_thresh=0.45, nms_topk=400, post_nms=-1)
        full = np.concatenate([t.asnumpy() for t in net(xin)], -1)          # (B, N*C, 6): box_nms output, un-sliced
        net.set_nms(nms_thresh=-1, nms_topk=400, post_nms=-1)
        prenms = np.concatenate([t.asnumpy() for t in net(xin)], -1)        # the detection tensor itself
        return table, params, [heads[i] for i in range(3)], prenms, full, (ids, scores, bboxes)

    def train(self, classes, x, gt_boxes, targets):
        mx = self.mx
        from mxnet import autograd, gluon
        net, struct, table, params = self.build(classes, x.shape[2])
        trainer = gluon.Trainer(net.collect_params(), 'sgd', {'learning_rate': 1e-3, 'wd': 5e-4, 'momentum': 0.9},
                                kvstore='local')                             # train_yolov3.py:527-530
        args = [mx.nd.array(a) for a in (x, gt_boxes) + tuple(targets)]
        with autograd.record():
            obj, ctr, scl, cls = net(*args)                                  # train_yolov3.py:625
            autograd.backward([obj + ctr + scl + cls])                       # :626,631
        losses = np.stack([t.asnumpy() for t in (obj, ctr, scl, cls)])
        grads = {k: p.grad().asnumpy() for k, p in struct.items() if p.grad_req != 'null' and k in params}
        running = {k: p.data().asnumpy() for k, p in struct.items() if "running" in k}
        trainer.step(x.shape[0])                                             # :634
        updated = {k: struct[k].data().asnumpy() for k in grads}
        return table, params, losses, grads, running, updated

    def save_params(self, scratch):
        mx = self.mx
        from mxnet.gluon import nn
        tiny = nn.HybridSequential()
        tiny.add(nn.Conv2D(4, 3, padding=1, use_bias=False, in_channels=3), nn.BatchNorm(in_channels=4),
                 nn.Conv2D(2, 1, in_channels=4))
        tiny.initialize(mx.init.Uniform(0.5))
        arrays = {k: v.data().asnumpy() for k, v in tiny._collect_params_with_prefix().items()}
        path = os.path.join(HERE, "mxnet_tiny.params")
        tiny.save_parameters(path)
        np.savez(os.path.join(HERE, "mxnet_tiny_params_expected.npz"), **arrays)
        # the full net: written to scratch, read back with this repo's reader, compared here
        net, struct, table, params = self.build(VOC, 64)
        big = os.path.join(scratch, "yolo3_darknet53_synthetic.params")
        net.save_parameters(big)
        mxp = _load_by_path("vy_mxparams", os.path.join(ROOT, "videoyolo_amd", "mxparams.py"))
        got = mxp.load(big)
        bad = [k for k, v in params.items() if k not in got or not np.array_equal(got[k], v)]
        print("mxparams.load on the full %d-array file: %s" % (len(got), "OK" if not bad else "MISMATCH %s" % bad[:5]))
        return not bad


class OracleBackend(object):
    """Plumbing check only (see the module docstring): the same captures from oracle/."""
    source = "oracle-selfcheck"

    def __init__(self, perturb=0.0):
        self.perturb = float(perturb)
        sys.path.insert(0, ROOT)
        from oracle import yolo3_oracle as O, yolo3_train_oracle as TO
        from videoyolo_amd import init
        self.O, self.TO, self.init = O, TO, init
        self.versions = "oracle/ of this repo, numpy %s" % np.__version__
        self.ops = KIT.OracleOps()

    def infer(self, classes, x, obj_bias=0.0):
        O = self.O
        C = len(classes)
        table = O.param_shapes(C)
        params = self.init.synthetic_params(table, seed=233, obj_bias=obj_bias)
        orc = O.OracleYolo3(C, params)
        heads = orc.raw_heads(x)
        if self.perturb:
            # stand-in for "another fp32 summation order" (what mxnet's MKL-DNN convolution is to this repo's): the heads
            # move by ~perturb relative, everything downstream is recomputed from them.  Shows what the first contact with
            # real goldens looks like: near-tie swaps that tests/test_mxnet_goldens.py must classify, not fail on
            rs = np.random.RandomState(99)
            heads = [(h * (1.0 + self.perturb * rs.standard_normal(h.shape))).astype(np.float32) for h in heads]
        prenms = orc.detections_from_heads(heads)
        full, _ = O.box_nms(prenms, 0.45, 0.01, 400)
        ids, scores, bboxes, _ = orc.nms(prenms)
        return table, params, heads, prenms, full, (ids, scores, bboxes)

    def train(self, classes, x, gt_boxes, targets):
        O, TO = self.O, self.TO
        C = len(classes)
        table = O.param_shapes(C)
        params = self.init.synthetic_params(table, seed=233)
        orc = TO.OracleYolo3Train(C, dict(params))
        losses = np.stack(orc.forward_train(x, gt_boxes, *targets))
        grads = orc.backward()
        running = dict(orc.new_running)
        upd = {k: v.copy() for k, v in params.items()}
        TO.sgd_step(upd, grads, {}, 1e-3, 0.9, 5e-4, x.shape[0])
        return table, params, losses, grads, running, {k: upd[k] for k in grads}

    def save_params(self, scratch):
        return True


# ------------------------------------------------------------------------------------------ capture
def meta(out, be, table, params, inputs):
    out["meta/source"] = np.array(be.source)
    out["meta/versions"] = np.array(be.versions)
    out["meta/param_names"] = np.array([k for k, _ in table])
    out["meta/param_shapes"] = np.array([",".join(map(str, s)) for _, s in table])
    out["meta/param_crc"] = np.array([crc(params[k]) for k, _ in table], np.uint32)
    for k, v in inputs.items():
        out["meta/crc_" + k] = np.array(crc(v), np.uint32)


def top_candidates(prenms, valid_thresh=0.01, n=TOP):
    """Per image the first n valid pre-NMS rows in box_nms order: (index (B, n) int64, scores (B, n) f32), -1 padded."""
    b = prenms.shape[0]
    idx = np.full((b, n), -1, np.int64)
    sc = np.full((b, n), -1, np.float32)
    for i in range(b):
        s_ = prenms[i, :, 1]
        v = np.nonzero(s_ > valid_thresh)[0]
        o = v[np.lexsort((v, -s_[v].astype(np.float64)))][:n]
        idx[i, :o.size], sc[i, :o.size] = o, s_[o]
    return idx, sc


def rows_to_index(prenms, rows):
    """For every non-filler row of `rows` (B, k, 6) the index of the bit-identical row of `prenms` (B, N, 6); -1 for
    filler, -2 where no row or more than one matches (identical candidates: the index is then not decidable from values)."""
    out = np.full(rows.shape[:2], -1, np.int64)
    for i in range(rows.shape[0]):
        table = {}
        cand = np.nonzero(prenms[i, :, 1] >= rows[i][rows[i, :, 0] >= 0][:, 1].min(initial=np.inf))[0]
        for r in cand:
            table.setdefault(prenms[i, r].tobytes(), []).append(int(r))
        for j in range(rows.shape[1]):
            if rows[i, j, 0] < 0:
                continue
            hit = table.get(np.ascontiguousarray(rows[i, j]).tobytes(), [])
            out[i, j] = hit[0] if len(hit) == 1 else -2
    return out


def capture_infer(be, outdir, size, seed, obj_bias=0.0, tag=""):
    x = frames(1, size, seed)
    table, params, heads, prenms, full, (ids, scores, bboxes) = be.infer(VOC, x, obj_bias)
    out = {}
    meta(out, be, table, params, {"x": x})
    out["in/size"], out["in/seed"] = np.array(size, np.int64), np.array(seed, np.int64)
    out["in/obj_bias"] = np.array(obj_bias, np.float32)
    for i, h in enumerate(heads):
        if size <= 416:
            out["head%d" % i] = np.asarray(h, np.float32)
        else:
            pack_sampled(out, "head%d" % i, h, 1000 + i)
    prenms = np.asarray(prenms, np.float32)
    pack_sampled(out, "prenms", prenms, 2000)
    out["top/index"], out["top/scores"] = top_candidates(prenms)
    full = np.asarray(full, np.float32)
    n_valid = int((full[0, :, 0] >= 0).sum())
    out["nms/first_rows"] = full[:, :max(400, n_valid)].copy()              # survivors are compacted to the front
    out["nms/first_rows_index"] = rows_to_index(prenms, out["nms/first_rows"])
    out["nms/rest_all_minus_one"] = np.array(bool((full[:, max(400, n_valid):] == -1).all()))
    out["nms/total_rows"] = np.array(full.shape[1], np.int64)
    out["ids"], out["scores"], out["bboxes"] = [np.asarray(t, np.float32) for t in (ids, scores, bboxes)]
    path = os.path.join(outdir, "mxnet_infer_%d%s.npz" % (size, tag))
    np.savez_compressed(path, **out)
    ts = out["top/scores"][0]
    gaps = np.abs(np.diff(ts[:401][ts[:401] >= 0].astype(np.float64)))
    print("wrote %s (%.1f KB): %d rows survive box_nms, %d returned; smallest gap among the top 401 pre-NMS scores %.3e" % (
        path, os.path.getsize(path) / 1e3, n_valid, int((ids >= 0).sum()), gaps.min() if gaps.size else float("nan")))


def capture_ops(be, outdir):
    """The operator-level kit: every case of mxnet_ops_kit.all_cases() through the backend's operators."""
    out = {"meta/source": np.array(be.source), "meta/versions": np.array(be.versions)}
    names = []
    for case in KIT.all_cases():
        res = be.ops.run(case["op"], case["inputs"], case["params"])
        pre = "ops/%s/" % case["name"]
        out[pre + "op"] = np.array(case["op"])
        out[pre + "decides"] = np.array(case["decides"])
        for k, v in case["inputs"].items():
            out[pre + "in/" + k] = np.asarray(v)
        for k, v in case["params"].items():
            out[pre + "par/" + k] = np.array(v, np.float64)
        for k, v in res.items():
            out[pre + "out/" + k] = np.asarray(v)
        names.append(case["name"])
    out["meta/cases"] = np.array(names)
    path = os.path.join(outdir, "mxnet_ops.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB): %d operator cases" % (path, os.path.getsize(path) / 1e3, len(names)))


def capture_train(be, outdir, size=96, b=2, seed=235):
    sys.path.insert(0, ROOT)
    tgt = _load_by_path("vy_targets", os.path.join(ROOT, "videoyolo_amd", "targets.py"))
    x = frames(b, size, seed)
    gt_boxes, gt_ids = synthetic_gt(b, size, len(VOC), m=3, seed=seed + 1)
    targets = tgt.YOLOV3PrefetchTargetGenerator(len(VOC))(size, size, gt_boxes, gt_ids)  # numpy in, numpy out
    table, params, losses, grads, running, updated = be.train(VOC, x, gt_boxes, targets)
    out = {}
    meta(out, be, table, params, {"x": x})
    out["in/size"], out["in/seed"], out["in/batch"] = np.array(size, np.int64), np.array(seed, np.int64), np.array(b, np.int64)
    out["in/gt_boxes"] = gt_boxes
    for name, t in zip(("obj_t", "centers_t", "scales_t", "weights_t", "clas_t"), targets):
        out["in/" + name] = np.asarray(t, np.float32)
    out["losses"] = np.asarray(losses, np.float32)                          # (4, B): obj, center, scale, cls
    for j, (k, g) in enumerate(sorted(grads.items())):
        pack_sampled(out, "grad/" + k, g, 3000 + j, n=1024)                 # 222 tensors: 1024 elements + sums each
        pack_sampled(out, "updated/" + k, updated[k], 3000 + j, n=1024)
    for k, v in running.items():
        out["running/" + k] = np.asarray(v, np.float32)
    path = os.path.join(outdir, "mxnet_train_%d.npz" % size)
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB): losses %s, %d gradient tensors" % (
        path, os.path.getsize(path) / 1e3, losses.sum(1).tolist(), len(grads)))


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--ref", default=os.environ.get("VIDEOYOLO_REF", "/root/reference"))
    ap.add_argument("--scratch", default="/tmp")
    ap.add_argument("--from-oracle", metavar="DIR", default=None)
    ap.add_argument("--perturb", type=float, default=0.0,
                    help="with --from-oracle: relative noise on the inference heads (1e-6 ~ another fp32 summation order) — "
                         "rehearses the near-tie adjudication of tests/test_mxnet_goldens.py")
    args = ap.parse_args()
    if args.from_oracle:
        be, outdir = OracleBackend(args.perturb), args.from_oracle
        os.makedirs(outdir, exist_ok=True)
        if os.path.realpath(outdir) == os.path.realpath(HERE):
            sys.exit("--from-oracle must not write into tests/golden/: those files would look like goldens")
    else:
        try:
            be = MxnetBackend(args.ref)
        except ImportError as e:
            sys.exit("needs mxnet + gluoncv + the reference tree at --ref (%s): %s" % (args.ref, e))
        outdir = HERE
    print("capturing with", be.versions)
    capture_ops(be, outdir)                 # the operator-level kit first: it is what localises a failure
    capture_infer(be, outdir, 416, 233)     # BASELINE configs[0]
    capture_infer(be, outdir, 416, 233, obj_bias=-7.0, tag="_sparse")   # ~50 valid candidates, wide score gaps: exact rows are a fair demand
    capture_infer(be, outdir, 608, 234)
    capture_train(be, outdir)
    if not be.save_params(args.scratch):
        sys.exit(1)


if __name__ == "__main__":
    main()
