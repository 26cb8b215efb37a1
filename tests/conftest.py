import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (before the oracle's first OpenMP region: on the GPU boxes the oracle's C kernels ran 20x
#               slower per call when libgomp was first initialised by them rather than after torch's own start-up —
#               `pytest tests/test_gpu_train_parity.py` alone took 5 minutes, the same tests 20 s in the full run)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


VOC_CLASSES = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
               'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train',
               'tvmonitor']  # /root/reference/datasets/names/pascalvoc.names


@pytest.fixture(scope="session")
def voc_classes():
    return list(VOC_CLASSES)


@pytest.fixture(scope="session")
def synth20():
    """Synthetic 20-class parameters (videoyolo_amd.init.synthetic_params, seed 233) as a dict of
    reference-layout numpy arrays: fed to BOTH the HIP path and the CPU oracle."""
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    return init.synthetic_params(O.param_shapes(20), seed=233)


def frames(batch, size, seed=233):
    """x ~ N(0,1), (B,3,S,S) fp32 — SURVEY §8d config 1 (seed 233 = train_yolov3.py:135)."""
    return np.random.default_rng(seed).standard_normal((batch, 3, size, size)).astype(np.float32)


def float64_cell_case():
    """Ground-truth boxes whose centres sit on stride multiples where the float64 value of
    `gtx / orig_width * width` (what yolo_target.py:115-116 computes under the reference's pinned
    NumPy 1.x: np.float32 scalar with Python ints) and its fp32 value truncate to DIFFERENT cells:
    at 416 px, x = 240 is cell 14 (float64: 14.999999999999998) but 15 in fp32 on the 26-wide map, and
    x = 120 / 240 are cells 14 / 29 vs 15 / 30 on the 52-wide map.  Returns (size, gt_boxes (1,M,4),
    gt_ids (1,M,1), expected) with expected = [(layer, anchor slot, cell_x, cell_y, tx, ty)] written out with
    plain Python floats — independent of the oracle."""
    size = 416
    rows = [
        # (x1, y1, x2, y2)            best zero-centred anchor       layer, slot
        ((225.0, 89.5, 255.0, 150.5), 1, 0),   # 30x61  -> (30,61): stride 16, slot 0; centre (240, 120)
        ((232.0, 105.0, 248.0, 135.0), 2, 1),  # 16x30  -> (16,30): stride 8, slot 1; centre (240, 120)
        ((115.0, 233.5, 125.0, 246.5), 2, 0),  # 10x13  -> (10,13): stride 8, slot 0; centre (120, 240)
    ]
    gt = np.full((1, len(rows) + 1, 4), -1, np.float32)
    ids = np.full((1, len(rows) + 1, 1), -1, np.float32)
    expected = []
    for m, (box, layer, slot) in enumerate(rows):
        gt[0, m] = box
        ids[0, m] = m
        w = size // (32, 16, 8)[layer]
        gx = float(np.float32(box[0]) + (np.float32(box[2]) - np.float32(box[0])) / np.float32(2))
        gy = float(np.float32(box[1]) + (np.float32(box[3]) - np.float32(box[1])) / np.float32(2))
        fx, fy = gx / size * w, gy / size * w          # Python floats = float64
        expected.append((layer, slot, int(fx), int(fy), np.float32(fx - int(fx)), np.float32(fy - int(fy))))
    return size, gt, ids, expected


def check_float64_cell_case(targets5, expected, size=416):
    """The rows `expected` names are the only positives, with the float64-derived centre targets."""
    obj, ctr = targets5[0], targets5[1]
    base, n = [], 0
    for s in (32, 16, 8):
        base.append(n)
        n += 3 * (size // s) ** 2
    want_rows = []
    for layer, slot, cx, cy, tx, ty in expected:
        w = size // (32, 16, 8)[layer]
        r = base[layer] + (cy * w + cx) * 3 + slot
        want_rows.append(r)
        assert obj[0, r, 0] == 1, (layer, slot, cx, cy)
        assert ctr[0, r, 0] == tx and ctr[0, r, 1] == ty, (ctr[0, r], tx, ty)
    assert sorted(np.nonzero(obj[0, :, 0])[0].tolist()) == sorted(want_rows)
