import os
import sys

import numpy as np
import pytest

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
