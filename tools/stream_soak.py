"""Soak of the pipelined host-fed detect loop (videoyolo_amd/stream.py): N batches alternating between a few distinct clip
batches, every result compared bit for bit with the synchronous path's result for that clip — pinned and pageable sources,
2 and 3 buffer slots.  A race between the copy / compute / copy-out streams would show as a mismatch sooner or later.

    python tools/stream_soak.py [--batches 300] [--batch 16] [--size 416]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import videoyolo_amd as vy  # noqa: E402
from videoyolo_amd import stream, transforms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=300)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--classes", type=int, default=30)
    args = ap.parse_args()
    import torch
    net = vy.yolo3_darknet53(["c%d" % i for i in range(args.classes)], pretrained_base=False)
    net.initialize(init="synthetic", seed=233, obj_bias=-3.0)
    net.collect_params().reset_ctx("cuda:0")
    net.set_nms(0.45, 400, 100)
    rng = np.random.default_rng(1)
    src_hw = (360, 640)
    clips = [rng.integers(0, 256, (args.batch,) + src_hw + (3,), dtype=np.uint8) for _ in range(3)]
    tf = transforms.YOLO3VideoInferenceTransform(args.size, args.size)
    want = [[t.cpu().numpy() for t in net(tf(c))] for c in clips]
    assert not np.array_equal(want[0][1], want[1][1])
    order = rng.integers(0, 3, args.batches)
    for depth in (2, 3):
        for pinned in (False, True):
            srcs = [torch.from_numpy(c).pin_memory() for c in clips] if pinned else clips
            det = stream.HostFedDetector(net, args.batch, src_hw, args.size, depth=depth)
            bad = 0
            t0 = time.perf_counter()
            for i, out in enumerate(det.run(srcs[j] for j in order)):
                w = want[order[i]]
                if not all(np.array_equal(a, b) for a, b in zip(out, w)):
                    bad += 1
            dt = time.perf_counter() - t0
            print("depth %d, %s source: %d batches of %d frames, %d mismatches, %.1f frames/s"
                  % (depth, "pinned" if pinned else "pageable", args.batches, args.batch, bad, args.batches * args.batch / dt))
            if bad:
                sys.exit(1)
    print("soak ok")


if __name__ == "__main__":
    main()
