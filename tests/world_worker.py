"""Worker of tests/test_world48.py: everything HOST-SIDE of the multi-GPU path under a real N-rank gloo group on CPU
(N = 4, 8 — the driver's scaling run is the first time the product meets more than 2 ranks; what can be checked without
N GPUs is checked here).  Reference call sites: train_yolov3.py:603-606 (split_and_load), :527-530 (kvstore reduce),
detect_yolo3.py:211-213 (even_split=False frame scatter) and :233 (result concat)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from videoyolo_amd import parallel  # noqa: E402


def main():
    parallel.init_process_group("gloo")
    dist = torch.distributed
    r, w = parallel.rank(), parallel.world_size()
    assert w == int(os.environ["WORLD_SIZE"]) and w in (4, 8)

    # ---- frame scatter with uneven clips (even_split=False: the first T % w ranks get one more), gather in rank order
    for total in (8 * w + 3, 4 * w, w + 1, 2 * w - 1):
        sizes = parallel.split_sizes(total, w)
        assert sum(sizes) == total and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
        clip = np.arange(total, dtype=np.float32)[:, None, None, None] * np.ones((1, 2, 2, 3), np.float32)
        mine = parallel.scatter_frames(clip)
        lo = sum(sizes[:r])
        assert mine.shape[0] == sizes[r] and (mine[:, 0, 0, 0] == np.arange(lo, lo + sizes[r])).all()
        # "detections" that carry the frame number: ids = frame, scores = frame / 1000, boxes = frame + [0, 1, 2, 3]
        f = torch.from_numpy(mine[:, 0, 0, 0].copy())
        ids = f[:, None, None].repeat(1, 5, 1)
        scores = ids / 1000.0
        bboxes = ids + torch.arange(4.0)
        for tot in (total, None):                      # with the total known (no size exchange) and without
            g_ids, g_sc, g_bb = parallel.gather_detections(ids, scores, bboxes, total=tot)
            assert g_ids.shape == (total, 5, 1) and g_bb.shape == (total, 5, 4)
            assert torch.equal(g_ids[:, 0, 0], torch.arange(float(total))), (total, tot, g_ids[:, 0, 0])
            assert torch.equal(g_bb[:, 2, 3], torch.arange(float(total)) + 3) and torch.allclose(g_sc, g_ids / 1000.0)

    # ---- control-plane agreement
    assert parallel.make_host_group() is not None and parallel.host_group() is not None
    assert parallel.any_rank(r == w - 1) and not parallel.any_rank(False) and parallel.any_rank(True)
    # a gloo SIDE group beside the default group (what an RCCL run creates): first the success path ...
    parallel._host_group = None
    g = parallel.make_host_group(side_group=True)
    assert g not in (None, False) and g is not dist.group.WORLD
    assert parallel.any_rank(r == 1) and not parallel.any_rank(False)
    # ... then ONE rank fails to create it: EVERY rank must end on the fallback (False), and agreement still works
    parallel._host_group = None
    real = parallel._new_gloo_group

    def flaky():
        grp = real()   # (creating a group is collective: every rank takes part, rank 2 then "loses" it)
        if r == 2:
            raise RuntimeError("no usable interface (simulated)")
        return grp
    parallel._new_gloo_group = flaky
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert parallel.make_host_group(side_group=True) is False
    parallel._new_gloo_group = real
    assert parallel.host_group() is False
    assert parallel.any_rank(r == 0) and not parallel.any_rank(False)
    parallel._host_group = None
    parallel.make_host_group()

    # ---- what the line of an N-rank run says about itself
    d = parallel.describe_group("cpu", extra={"streamk_enabled": None})
    assert d["backend"] == "gloo" and d["world"] == w and d["ranks_in_order"] and len(d["devices"]) == w
    assert [x["rank"] for x in d["devices"]] == list(range(w)) and len({x["pid"] for x in d["devices"]}) == w
    assert d["ipc_mode"] == "dmabuf"                   # the launcher sets HSA_ENABLE_IPC_MODE_LEGACY=0 for every rank

    # ---- pre-flight: passes, and fails THE SAME WAY on every rank
    pf = parallel.preflight("cpu", nbytes=1 << 20, iters=3)
    assert pf["ok"] and pf["world"] == w and pf["allreduce_bytes"] == 1 << 20 and pf["allreduce_busbw_GBps"] > 0
    bad = parallel.preflight("cpu", nbytes=1 << 16, iters=2, min_busbw_GBps=1e9)
    assert not bad["ok"] and "bus bandwidth" in bad["reason"]
    reasons = [None] * w
    dist.all_gather_object(reasons, bad["reason"])
    assert len(set(reasons)) == 1, reasons

    # ---- gradient all-reduce over the bucket table: the host-side table covers every trainable tensor exactly once, is the
    # same on every rank, and reducing bucket by bucket equals one all-reduce of the whole buffer
    import videoyolo_amd as vy
    net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False)     # (host-only: no device is touched)
    table = parallel.grad_bucket_table(net)
    assert [t[0] for t in table] == ["heads", "stages.2", "stages.1", "stages.0"]
    spans = sorted((o, o + c) for _, o, c in table)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), spans
    for p in net.collect_params().values():
        inside = [s for s in spans if s[0] <= p.offset and p.offset + p.size <= s[1]]
        assert len(inside) == (1 if p.trainable else len(inside)) and (inside or not p.trainable), p.name
    tables = [None] * w
    dist.all_gather_object(tables, table)
    assert all(t == table for t in tables)
    n = int(net._lib.vy_net_param_bytes(net._h) // 4)
    rs = np.random.RandomState(r)
    grads = torch.from_numpy(rs.randint(-8, 8, n).astype(np.float32))     # small integers: sums are exact in any order
    whole = grads.clone()
    parallel.allreduce_(whole)
    for _, o, c in table:
        parallel.allreduce_(grads[o:o + c])
    for _, o, c in table:
        assert torch.equal(grads[o:o + c], whole[o:o + c])
    dist.barrier()
    print("rank %d of %d ok" % (r, w))


if __name__ == "__main__":
    main()
