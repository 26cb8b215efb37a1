"""The steps either side of the model call in the reference's inference driver.

  before  YOLO3VideoInferenceTransform.__call__  models/definitions/yolo/transforms.py:316-350
          (resize -> to_tensor -> normalize) as ONE HIP kernel (csrc/preproc.hip): imresize(interp=9) =
          OpenCV area (shrink) / bicubic (enlarge) / bilinear (mixed) on the uint8 frame, restated from memory
          (no OpenCV offline: the CPU checker restates every rounding step and is cross-checked against torch / exact area definitions),
          fused with to_tensor + normalize; frames already at the network size skip the resize.
  after   detect_yolo3.py:226 (clip to the image), :256-265 (drop id < 0 rows, boxes / image size,
          one [id, score, x1, y1, x2, y2] row per detection), :327-330 (the prediction txt line).
"""
import ctypes

import numpy as np

from . import _lib

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


class YOLO3VideoInferenceTransform(object):
    def __init__(self, width, height, mean=MEAN, std=STD):
        self._width, self._height = width, height
        self._mean = np.asarray(mean, np.float32)
        self._std = np.asarray(std, np.float32)

    def __call__(self, frames, device="cuda:0"):
        """frames: (B,h,w,3) or (h,w,3) uint8 (numpy or torch), any size -> (B,3,height,width) fp32 normalised
        torch tensor on `device` (resized like timage.imresize(frame, width, height, interp=9))."""
        import torch
        lib = _lib.load()
        x = frames if isinstance(frames, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(frames))
        if x.dtype != torch.uint8:
            raise TypeError("frames must be uint8 (decoded images), got %s" % x.dtype)
        if x.dim() == 3:
            x = x[None]
        b, h, w, c = x.shape
        if c != 3:
            raise ValueError("expected (B,h,w,3) frames, got %s" % (tuple(x.shape),))
        x = x.to(device).contiguous()
        out = torch.empty((b, 3, self._height, self._width), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.vy_preprocess_resize_frames(
                ctypes.c_void_p(x.data_ptr()), h, w, ctypes.c_void_p(out.data_ptr()), b, self._height, self._width,
                self._mean.ctypes.data_as(ctypes.c_void_p), self._std.ctypes.data_as(ctypes.c_void_p),
                ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return out


def postprocess(ids, scores, bboxes, size):
    """detect_yolo3.py:226,256-265 for one batch: clip boxes to [0, size], keep rows with id >= 0,
    normalise boxes by size.  Returns a list (per image) of float arrays (k, 6): id, score, x1, y1, x2, y2."""
    to_np = lambda t: t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    ids, scores, bboxes = to_np(ids), to_np(scores), to_np(bboxes)
    bboxes = np.clip(bboxes, 0, size)
    out = []
    for i in range(ids.shape[0]):
        valid = np.where(ids[i].flat >= 0)[0]
        box = bboxes[i][valid, :] / size
        out.append(np.concatenate([ids[i].flat[valid].astype(int)[:, None].astype(np.float64),
                                   scores[i].flat[valid][:, None].astype(np.float64),
                                   box.astype(np.float64)], axis=1))
    return out


def prediction_lines(img_path, rows):
    """The reference's prediction file format (detect_yolo3.py:327-330):
    ``path,class,score,x1,y1,x2,y2`` per detection, class printed as an int."""
    return ["{},{},{},{},{},{},{}\n".format(img_path, int(r[0]), r[1], r[2], r[3], r[4], r[5]) for r in rows]
