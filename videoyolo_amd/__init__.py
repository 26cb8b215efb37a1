"""videoyolo_amd — MI355X-native yolo3_darknet53 hot path (hand-written HIP for gfx950 behind a
C-ABI, include/vyolo.h) with the model-object surface of HaydenFaulkner/VideoYOLO's
``models.definitions.yolo.wrappers.yolo3_darknet53``."""
from . import autograd  # noqa: F401
from .model import YOLOV3, YOLOV3T, BatchNorm, SyncBatchNorm, yolo3_darknet53  # noqa: F401
from .trainer import Trainer  # noqa: F401
from . import parallel  # noqa: F401
from . import lr_scheduler  # noqa: F401
from .lr_scheduler import LRScheduler, LRSequential  # noqa: F401

__all__ = ["yolo3_darknet53", "YOLOV3", "YOLOV3T", "BatchNorm", "SyncBatchNorm", "autograd", "Trainer", "parallel", "lr_scheduler",
           "LRScheduler", "LRSequential"]
