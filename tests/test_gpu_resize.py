"""-m gpu: the fused resize + to_tensor + normalize kernel (csrc/preproc.hip through vy_preprocess_resize_frames)
against the restated reference transform (oracle/resize_oracle.py: YOLO3VideoInferenceTransform.__call__,
models/definitions/yolo/transforms.py:316-350 with imresize(interp=9) -> OpenCV area / cubic / linear on uint8).
Integer arithmetic and the rounding to uint8 must be bit-exact; the final float values too (same fp32 operation
sequence, no contraction)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(b, h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = 127 + 100 * np.sin(x / 7.0)[..., None] * np.cos(y / 5.0)[..., None] * np.array([1, 0.5, -1])
    return np.clip(base[None] + rng.normal(0, 25, (b, h, w, 3)), 0, 255).astype(np.uint8)


CASES = [
    ("cubic enlarge", (2, 120, 160), (416, 416)),
    ("cubic enlarge odd", (1, 37, 53), (64, 96)),
    ("area fractional (720p -> 608)", (2, 720, 1280), (608, 608)),
    ("area fractional (480p -> 416)", (1, 480, 640), (416, 416)),
    ("area 2x2", (2, 192, 256), (96, 128)),
    ("area 3x3", (1, 288, 384), (96, 128)),
    ("area 4x2", (1, 256, 512), (128, 128)),
    ("linear mixed", (2, 50, 200), (96, 96)),
    ("linear one side equal", (1, 96, 300), (96, 128)),
    ("same size", (2, 96, 96), (96, 96)),
]


@pytest.mark.parametrize("name,src,dst", CASES, ids=[c[0] for c in CASES])
def test_resize_normalize_matches_the_restated_transform(name, src, dst):
    from videoyolo_amd import transforms
    from oracle import resize_oracle as R
    frames = _frames(*src, seed=len(name))
    t = transforms.YOLO3VideoInferenceTransform(dst[1], dst[0])       # (width, height) like the reference
    got = t(frames).cpu().numpy()
    want, resized = R.inference_transform(frames, dst[1], dst[0])
    assert got.shape == want.shape == (src[0], 3) + dst
    # recover the uint8 value the kernel rounded to and compare the integers first (clearer failure)
    mean, std = np.asarray(transforms.MEAN, np.float32), np.asarray(transforms.STD, np.float32)
    back = np.rint((got.transpose(0, 2, 3, 1) * std + mean) * 255.0).astype(np.int64)
    diff = np.abs(back - resized.astype(np.int64))
    assert diff.max() == 0, "%s: %d pixels differ, max %d levels" % (name, int((diff > 0).sum()), int(diff.max()))
    assert np.array_equal(got, want)


def test_resized_frames_feed_the_network(voc_classes, synth20):
    """The transform's output is the network input: 720p frames -> 416 x 416 -> detections, identical to feeding the
    oracle-resized frames."""
    import videoyolo_amd as vy
    from videoyolo_amd import transforms
    from oracle import resize_oracle as R
    frames = _frames(2, 360, 640, seed=3)
    net = vy.yolo3_darknet53(voc_classes, pretrained_base=False)
    net.set_parameters(synth20)
    net.collect_params().reset_ctx("cuda:0")
    x = transforms.YOLO3VideoInferenceTransform(224, 224)(frames)
    want, _ = R.inference_transform(frames, 224, 224)
    a = [t.cpu().numpy() for t in net(x, return_index=True)]
    b = [t.cpu().numpy() for t in net(want, return_index=True)]
    assert all(np.array_equal(p, q) for p, q in zip(a, b))


def test_extreme_shrink_is_refused():
    from videoyolo_amd import _lib, transforms
    with pytest.raises(_lib.VyError):
        transforms.YOLO3VideoInferenceTransform(32, 32)(np.zeros((1, 500, 700, 3), np.uint8))
