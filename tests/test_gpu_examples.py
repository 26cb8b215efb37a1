"""-m gpu: the two example scripts (examples/detect.py, examples/train.py) run end to end — the reference's inference
and training loops (detect_yolo3.py:199-330, train_yolov3.py:494-640) driven through the public surface only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=timeout, universal_newlines=True)
    assert p.returncode == 0, p.stdout[-3000:]
    return p.stdout


def test_detect_example():
    out = _run(["examples/detect.py", "--size", "416", "--batch", "2", "--frames", "3"])
    assert "3 frames" in out and "prediction lines" in out and "mAP" in out


def test_detect_example_pipelined_equals_synchronous_and_two_ranks():
    """examples/detect.py is the pipelined host-fed loop (videoyolo_amd/stream.py); --sync is the reference's shape (transform,
    net(x), .cpu(), one after the other); --gpus 2 scatters every clip batch over two ranks (sharing the test box's GPU over
    gloo) and gathers the rows to rank 0.  All three print the same prediction lines and the same mAP."""
    a = _run(["examples/detect.py", "--size", "416", "--batch", "4", "--frames", "10"])
    b = _run(["examples/detect.py", "--size", "416", "--batch", "4", "--frames", "10", "--sync"])
    c = _run(["examples/detect.py", "--size", "416", "--batch", "4", "--frames", "10", "--gpus", "2", "--backend", "gloo", "--share-gpu"])
    pick = lambda o: [l for l in o.splitlines() if "prediction lines" in l or "mAP" in l]  # noqa: E731
    assert len(pick(a)) == 2 and pick(a) == pick(b) == pick(c)


def test_detect_example_in_the_split_conv_mode():
    exact = _run(["examples/detect.py", "--size", "416", "--batch", "2", "--frames", "3"])
    split = _run(["examples/detect.py", "--size", "416", "--batch", "2", "--frames", "3", "--conv-mode", "split_bf16x3"])
    assert "3 frames" in split and "mAP" in split
    # the same detections survive (synthetic weights with a low objectness bias: a handful of boxes)
    assert exact.split("prediction lines")[0] == split.split("prediction lines")[0]


@pytest.mark.parametrize("mode", ["exact", "split_bf16x3_train"])
def test_train_example(tmp_path, mode):
    out = _run(["examples/train.py", "--batch", "2", "--size", "96", "--steps", "3", "--conv-mode", mode])
    steps = [l for l in out.splitlines() if l.startswith("step ")]
    assert len(steps) == 3 and "saved" in out
    vals = [float(l.split("obj")[1].split()[0]) for l in steps]
    assert all(v == v and v > 0 for v in vals)
