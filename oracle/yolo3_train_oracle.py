"""CPU oracle for the TRAINING branch of the yolo3_darknet53 hot path.

TEST INFRASTRUCTURE ONLY (see yolo3_oracle.py).  PARITY UNPINNED: BatchNorm (train), YOLOV3Loss,
BBoxBatchIOU, autograd and the SGD update live in mxnet / gluoncv; what is restated here is their
published definition at the reference's call sites, each tagged [UPSTREAM-RECALLED] where the
detail is not visible in /root/reference.  Gradients are derived by hand from those definitions and
cross-checked against torch.autograd in tests/test_oracle_train_vs_torch.py.

Reference call sites (relative to /root/reference):
  train-mode forward     models/definitions/yolo/yolo3.py:1126-1187 (autograd.is_recording branch)
  raw predictions        models/definitions/yolo/yolo3.py:158-182   (train returns raw xy/wh/obj/cls)
  dynamic targets        models/definitions/yolo/yolo_target.py:173-205
  target merge           models/definitions/yolo/yolo_target.py:226-281
  loss                   gluoncv.loss.YOLOV3Loss, constructed at yolo3.py:994, called at :1187
  backward + update      train_yolov3.py:626-634 (sum of the 4 losses, autograd.backward,
                         trainer.step(batch_size)); Trainer('sgd', wd, momentum) :527-530
"""
import ctypes

import numpy as np

from . import yolo3_oracle as O

_f32p = ctypes.POINTER(ctypes.c_float)


def _p(a):
    return a.ctypes.data_as(_f32p)


def conv_bwd_data(dy, w, stride, pad, in_hw):
    dy, w = O._c(dy), O._c(w)
    n, o, ho, wo = dy.shape
    _, c, k, _ = w.shape
    h, wd = in_hw
    dx = np.empty((n, c, h, wd), np.float32)
    O.lib().vyo_conv2d_bwd_data(_p(dy), n, o, ho, wo, _p(w), c, k, stride, pad, h, wd, _p(dx))
    return dx


def conv_bwd_weight(dy, x, k, stride, pad):
    dy, x = O._c(dy), O._c(x)
    n, o, ho, wo = dy.shape
    _, c, h, wd = x.shape
    dw = np.empty((o, c, k, k), np.float32)
    O.lib().vyo_conv2d_bwd_weight(_p(dy), n, o, ho, wo, _p(x), c, k, stride, pad, h, wd, _p(dw))
    return dw


def sigmoid_bce(x, z):
    """gluon SigmoidBinaryCrossEntropyLoss(from_sigmoid=False) elementwise term [UPSTREAM-RECALLED]:
    relu(x) - x*z + softrelu(-|x|), softrelu(t) = log(1 + exp(t))."""
    x = x.astype(np.float32)
    return (np.maximum(x, 0) - x * z + O.log(np.float32(1.0) + O.exp(-np.abs(x)))).astype(np.float32)


class OracleYolo3Train(O.OracleYolo3):
    BN_MOMENTUM = 0.9  # layers.py:68
    BN_EPS = 1e-5
    RUNNING_VAR_UNBIASED = False  # [UPSTREAM-RECALLED] mxnet CPU BatchNorm keeps the biased variance

    def __init__(self, num_class, params, ignore_iou_thresh=0.7, label_smooth=False, device_slices=None,
                 sync_bn=False):
        """device_slices: data parallelism as the reference runs it (split_and_load, train_yolov3.py:603-606) —
        a list of batch-axis slices, one per device.  Every BatchNorm then normalises with the statistics of
        ITS device's slice, except — with sync_bn (SyncBatchNorm(num_devices), train_yolov3.py:352-354) — the
        layers that are actually built from the passed norm_layer: the stem and the five stride-2 convs of
        Darknet-53 (three_darknet.py:163-181; the residual blocks hard-code BatchNorm :193-194 and
        wrappers.py:101-103 does not forward norm_layer to the heads), which use the whole batch.  Gradients
        are the sums over all devices (Trainer kvstore reduce)."""
        super().__init__(num_class, params)
        self.ignore_iou_thresh = ignore_iou_thresh
        self.label_smooth = label_smooth
        self.device_slices = device_slices
        self.sync_bn = sync_bn
        self.tape = []
        self.new_running = {}
        self.new_running_dev = []

    def _stat_groups(self, pre, batch):
        """batch-axis slices over which the BatchNorm of cell `pre` takes its statistics"""
        whole = [slice(0, batch)]
        if not self.device_slices:
            return whole
        synced = self.sync_bn and pre.startswith("stages.") and ".body." not in pre
        return whole if synced else list(self.device_slices)

    # ---------------------------------------------------------------- forward (recording)
    def cell(self, x, pre, k, stride):
        """_conv2d cell in train mode: batch statistics (layers.py:63-70 under autograd.record)."""
        w = self.p[pre + ".0.weight"]
        z = O.conv2d(x, w, stride, k // 2)
        g, b = self.p[pre + ".1.gamma"], self.p[pre + ".1.beta"]
        groups = self._stat_groups(pre, z.shape[0])
        y = np.empty_like(z)
        means, vars_ = [], []
        m = np.float32(self.BN_MOMENTUM)
        ndev = len(self.device_slices) if self.device_slices else 1
        while len(self.new_running_dev) < ndev:
            self.new_running_dev.append({})
        for gi, sl in enumerate(groups):
            y[sl], mean, var = O.bn_train(z[sl], g, b, self.BN_EPS, leaky=True)
            n = z[sl].shape[0] * z.shape[2] * z.shape[3]
            rv = var * (n / max(n - 1, 1)) if self.RUNNING_VAR_UNBIASED else var
            new = {pre + ".1.running_mean": self.p[pre + ".1.running_mean"] * m + mean * (1 - m),
                   pre + ".1.running_var": self.p[pre + ".1.running_var"] * m + rv * (1 - m)}
            # every device keeps its own running statistics; a layer normalised over the whole batch
            # gives all devices the same ones
            for d in (range(ndev) if len(groups) == 1 else [gi]):
                self.new_running_dev[d].update(new)
            means.append(mean)
            vars_.append(var)
        self.new_running.update({k: v for k, v in self.new_running_dev[0].items() if k.startswith(pre + ".1.")})
        self.tape.append(dict(kind="cell", pre=pre, k=k, s=stride, x=x, z=z, mean=means, var=vars_, groups=groups,
                              out=y))
        return y

    def forward_raw(self, x):
        """Runs stages + heads in train mode.  Returns per-scale prediction conv outputs and keeps
        the graph on self.tape (a DAG walked in reverse by backward())."""
        self.tape, self.new_running, self.new_running_dev = [], {}, []
        x = O._c(x)
        feats = O.darknet_feature_cells()
        routes = []
        self.acts = {}
        for si, (lo, hi) in enumerate(O.STAGE_SLICES):
            for j, f in enumerate(feats[lo:hi]):
                pre = "stages.%d.%d" % (si, j)
                if f[0] == "conv":
                    x = self.cell(x, pre, f[3], f[4])
                else:
                    r = x
                    y = self.cell(x, pre + ".body.0", 1, 1)
                    y = self.cell(y, pre + ".body.1", 3, 1)
                    x = y + r
                    self.tape.append(dict(kind="add"))
            routes.append(x)
            self.tape.append(dict(kind="route", idx=si))
        preds = []
        x = routes[-1]
        for i in range(3):
            for j in range(5):
                x = self.cell(x, "yolo_blocks.%d.body.%d" % (i, j), 1 if j % 2 == 0 else 3, 1)
            route = x
            tip = self.cell(route, "yolo_blocks.%d.tip" % i, 3, 1)
            w = self.p["yolo_outputs.%d.prediction.weight" % i]
            pred = O.conv2d(tip, w, 1, 0, None, self.p["yolo_outputs.%d.prediction.bias" % i])
            self.tape.append(dict(kind="pred", i=i, x=tip))
            preds.append(pred)
            if i == 2:
                break
            t = self.cell(route, "transitions.%d" % i, 1, 1)
            up = t.repeat(2, axis=-1).repeat(2, axis=-2)
            route_now = routes[::-1][i + 1]
            x = np.concatenate([up, route_now], axis=1)
            self.tape.append(dict(kind="cat", c_up=up.shape[1]))
        return preds

    # yolo3.py:158-182 + 1143-1148: per-scale raw tensors in (B, HW*A, .) order, and decoded boxes
    def split_preds(self, preds):
        C, A = self.C, 3
        P = 5 + C
        outs = dict(xy=[], wh=[], obj=[], cls=[], box=[])
        for i, pred in enumerate(preds):
            B, _, H, W = pred.shape
            pr = pred.reshape(B, A * P, H * W).transpose(0, 2, 1).reshape(B, H * W, A, P)
            raw_xy, raw_wh, obj, cls = pr[..., 0:2], pr[..., 2:4], pr[..., 4:5], pr[..., 5:]
            anchors = np.array(O.ANCHORS[::-1][i], np.float32).reshape(1, 1, A, 2)
            stride = np.float32(O.STRIDES[::-1][i])
            gx, gy = np.meshgrid(np.arange(W), np.arange(H))
            offsets = np.stack([gx, gy], -1).astype(np.float32).reshape(1, H * W, 1, 2)
            centers = (O.sigmoid(raw_xy) + offsets) * stride
            scales = O.exp(raw_wh) * anchors
            wh = scales / np.float32(2.0)
            bbox = np.concatenate([centers - wh, centers + wh], -1)
            outs["xy"].append(raw_xy.reshape(B, -1, 2))
            outs["wh"].append(raw_wh.reshape(B, -1, 2))
            outs["obj"].append(obj.reshape(B, -1, 1))
            outs["cls"].append(cls.reshape(B, -1, C))
            outs["box"].append(bbox.reshape(B, -1, 4))
        return {k: np.concatenate(v, 1).astype(np.float32) for k, v in outs.items()}

    # yolo_target.py:173-205 + 226-281
    def merge_targets(self, box_preds, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t):
        ious = O.batch_iou(box_preds, gt_boxes)  # (B,N,M)
        ious_max = ious.max(axis=-1, keepdims=True)
        dyn_obj = (ious_max > np.float32(self.ignore_iou_thresh)).astype(np.float32) * -1  # :204
        mask = obj_t > 0  # :263
        objness = np.where(mask, obj_t, dyn_obj)
        center_t = np.where(mask, centers_t, 0).astype(np.float32)
        scale_t = np.where(mask, scales_t, 0).astype(np.float32)
        weight_t = np.where(mask, weights_t, 0).astype(np.float32)
        class_t = np.where(mask, clas_t, -1).astype(np.float32)
        if self.label_smooth:  # :272-278
            sw = np.float32(min(1.0 / self.C, 1.0 / 40))
            class_t = np.where(class_t > 0.5, class_t - sw, class_t)
            class_t = np.where((class_t < -0.5) | (class_t > 0.5), class_t, sw).astype(np.float32)
        class_mask = mask * (class_t >= 0)
        return (objness.astype(np.float32), center_t, scale_t, weight_t, class_t,
                class_mask.astype(np.float32))

    # gluoncv.loss.YOLOV3Loss [UPSTREAM-RECALLED]; every loss is mean(...) * count == a per-sample sum
    def loss(self, pr, tg):
        objness, center_t, scale_t, weight_t, class_t, class_mask = tg
        weight_t = weight_t * objness
        hard = np.where(objness > 0, 1.0, objness).astype(np.float32)
        omask = np.where(objness > 0, objness, (objness >= 0).astype(np.float32)).astype(np.float32)
        cmask = class_mask * objness
        f64 = np.float64
        obj_l = (sigmoid_bce(pr["obj"], hard) * omask).astype(f64).sum(axis=(1, 2))
        ctr_l = (sigmoid_bce(pr["xy"], center_t) * weight_t).astype(f64).sum(axis=(1, 2))
        scl_l = (np.abs(pr["wh"] - scale_t) * weight_t).astype(f64).sum(axis=(1, 2))
        cls_l = (sigmoid_bce(pr["cls"], class_t) * cmask).astype(f64).sum(axis=(1, 2))
        # d(sum of the four, unit head gradients)/d(raw predictions)
        g_obj = (O.sigmoid(pr["obj"]) - hard) * omask
        g_xy = (O.sigmoid(pr["xy"]) - center_t) * weight_t
        g_wh = np.sign(pr["wh"] - scale_t) * weight_t
        g_cls = (O.sigmoid(pr["cls"]) - class_t) * cmask
        grads = dict(obj=g_obj.astype(np.float32), xy=g_xy.astype(np.float32), wh=g_wh.astype(np.float32),
                     cls=g_cls.astype(np.float32))
        return [l.astype(np.float32) for l in (obj_l, ctr_l, scl_l, cls_l)], grads

    def forward_train(self, x, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t):
        """net(x, gt_boxes, *targets) under autograd.record(): the four (B,) losses."""
        preds = self.forward_raw(x)
        pr = self.split_preds(preds)
        tg = self.merge_targets(pr["box"], gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t)
        losses, g = self.loss(pr, tg)
        # scatter the raw-prediction gradients back to the (B, A*P, H, W) layout of each head
        C, A = self.C, 3
        P = 5 + C
        dpreds, n0 = [], 0
        for pred in preds:
            B, _, H, W = pred.shape
            n1 = n0 + H * W * A
            d = np.concatenate([g["xy"][:, n0:n1], g["wh"][:, n0:n1], g["obj"][:, n0:n1], g["cls"][:, n0:n1]], -1)
            d = d.reshape(B, H * W, A * P).transpose(0, 2, 1).reshape(B, A * P, H, W)
            dpreds.append(np.ascontiguousarray(d, np.float32))
            n0 = n1
        self._dpreds, self._targets, self._split = dpreds, tg, pr
        return losses

    # ---------------------------------------------------------------- backward
    def _cell_bwd(self, t, da, grads):
        pre = t["pre"]
        g = self.p[pre + ".1.gamma"].astype(np.float64).reshape(1, -1, 1, 1)
        b = self.p[pre + ".1.beta"].astype(np.float64).reshape(1, -1, 1, 1)
        dz = np.empty(t["z"].shape, np.float32)
        dbeta_tot, dgamma_tot = 0.0, 0.0
        for sl, mean, var in zip(t["groups"], t["mean"], t["var"]):  # one statistics group per device (or one)
            z = t["z"][sl].astype(np.float64)
            mean = mean.astype(np.float64).reshape(1, -1, 1, 1)
            inv = 1.0 / np.sqrt(var.astype(np.float64).reshape(1, -1, 1, 1) + self.BN_EPS)
            xhat = (z - mean) * inv
            y = xhat * g + b
            dy = da[sl].astype(np.float64) * np.where(y > 0, 1.0, 0.1)
            n = z.shape[0] * z.shape[2] * z.shape[3]
            dbeta = dy.sum(axis=(0, 2, 3))
            dgamma = (dy * xhat).sum(axis=(0, 2, 3))
            dz[sl] = (g * inv * (dy - dbeta.reshape(1, -1, 1, 1) / n - xhat * dgamma.reshape(1, -1, 1, 1) / n)
                      ).astype(np.float32)
            dbeta_tot, dgamma_tot = dbeta_tot + dbeta, dgamma_tot + dgamma
        grads[pre + ".1.gamma"] = np.asarray(dgamma_tot).astype(np.float32)
        grads[pre + ".1.beta"] = np.asarray(dbeta_tot).astype(np.float32)
        w = self.p[pre + ".0.weight"]
        grads[pre + ".0.weight"] = conv_bwd_weight(dz, t["x"], t["k"], t["s"], t["k"] // 2)
        if t["x"].shape[1] == 3:
            return None  # image gradient is not needed
        return conv_bwd_data(dz, w, t["s"], t["k"] // 2, t["x"].shape[2:])

    def backward(self):
        """autograd.backward(sum of the four losses) (train_yolov3.py:626-631): returns
        {structural name: gradient} in reference layouts."""
        grads = {}
        tape = list(self.tape)
        route_grads = [None, None, None]
        # ---- heads, shallow -> deep in reverse of the forward order
        # replay structure explicitly (mirrors forward_raw)
        cells = {t["pre"]: t for t in tape if t.get("kind") == "cell"}
        preds = {t["i"]: t for t in tape if t.get("kind") == "pred"}
        g_x = None  # gradient flowing into the head input of scale i (from scale i+1's concat)
        head_in_grad = [None, None, None]
        for i in (2, 1, 0):
            dp = self._dpreds[i]
            tip = preds[i]["x"]
            wname = "yolo_outputs.%d.prediction" % i
            grads[wname + ".bias"] = dp.astype(np.float64).sum(axis=(0, 2, 3)).astype(np.float32)
            grads[wname + ".weight"] = conv_bwd_weight(dp, tip, 1, 1, 0)
            d_tip = conv_bwd_data(dp, self.p[wname + ".weight"], 1, 0, tip.shape[2:])
            d_route = self._cell_bwd(cells["yolo_blocks.%d.tip" % i], d_tip, grads)
            if i < 2:
                # the transition of scale i consumed `route` too; its gradient arrives from scale i+1
                d_up = head_in_grad[i + 1][:, :self._c_up(i)]
                B, c, H2, W2 = d_up.shape
                d_t = d_up.reshape(B, c, H2 // 2, 2, W2 // 2, 2).astype(np.float64).sum(axis=(3, 5)).astype(np.float32)
                d_route = d_route + self._cell_bwd(cells["transitions.%d" % i], d_t, grads)
                route_grads[1 - i] = head_in_grad[i + 1][:, self._c_up(i):]
            d = d_route
            for j in (4, 3, 2, 1, 0):
                d = self._cell_bwd(cells["yolo_blocks.%d.body.%d" % (i, j)], d, grads)
            head_in_grad[i] = d
        route_grads[2] = head_in_grad[0]
        # ---- backbone, deep -> shallow
        feats = O.darknet_feature_cells()
        d = None
        for si in (2, 1, 0):
            lo, hi = O.STAGE_SLICES[si]
            d = route_grads[si] if d is None else d + route_grads[si]
            for j in range(hi - lo - 1, -1, -1):
                f = feats[lo + j]
                pre = "stages.%d.%d" % (si, j)
                if f[0] == "conv":
                    d = self._cell_bwd(cells[pre], d, grads)
                else:
                    db = self._cell_bwd(cells[pre + ".body.1"], d, grads)
                    db = self._cell_bwd(cells[pre + ".body.0"], db, grads)
                    d = d + db
        return grads

    def _c_up(self, i):
        return O.HEAD_CHANNELS[i] // 2


def sgd_step(params, grads, mom, lr, momentum, wd, batch_size, wd_mult=None):
    """mx.optimizer.SGD as driven by gluon.Trainer.step(batch_size) [UPSTREAM-RECALLED]:
    g = grad / batch_size ; mom = momentum*mom - lr*(g + wd*w) ; w += mom."""
    for k, g in grads.items():
        w = params[k]
        wdk = wd * (wd_mult.get(k, 1.0) if wd_mult else 1.0)
        g = g * np.float32(1.0 / batch_size)
        m = mom.get(k, np.zeros_like(w))
        m = np.float32(momentum) * m - np.float32(lr) * (g + np.float32(wdk) * w)
        mom[k] = m.astype(np.float32)
        params[k] = (w + m).astype(np.float32)
