"""videoyolo_amd — MI355X-native yolo3_darknet53 hot path (hand-written HIP for gfx950 behind a
C-ABI, include/vyolo.h) with the model-object surface of HaydenFaulkner/VideoYOLO's
``models.definitions.yolo.wrappers.yolo3_darknet53``."""
import os as _os

# dmabuf IPC: RCCL (and CUDA-tensor sharing) across processes fails with `hipIpcGetMemHandle: invalid argument` on hosts
# whose driver only supports it unless the HSA runtime is told so BEFORE it initialises, i.e. before the first GPU call of
# the process.  Set here — package import precedes every GPU call the package makes — so that a rank started by ANY
# launcher (torch.distributed.run, videoyolo_amd.launch, mpirun) has it, not only ranks started by the repo's own launcher.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import autograd  # noqa: F401,E402
from .model import YOLOV3, YOLOV3T, BatchNorm, SyncBatchNorm, yolo3_darknet53  # noqa: F401,E402
from .trainer import Trainer  # noqa: F401,E402
from . import parallel  # noqa: F401,E402
from . import lr_scheduler  # noqa: F401,E402
from .lr_scheduler import LRScheduler, LRSequential  # noqa: F401,E402

__all__ = ["yolo3_darknet53", "YOLOV3", "YOLOV3T", "BatchNorm", "SyncBatchNorm", "autograd", "Trainer", "parallel", "lr_scheduler",
           "LRScheduler", "LRSequential"]
