"""Small-batch inference latency, eager vs hybridized (HIP graph replay)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import videoyolo_amd as vy
net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False)
net.initialize(init="synthetic", seed=233, obj_bias=-4.0)
net.collect_params().reset_ctx("cuda:0")
for size in (416, 608):
    for b in (1, 4):
        x = torch.randn((b, 3, size, size), device="cuda:0")
        res = {}
        for mode in ("eager", "graph"):
            net.hybridize(mode == "graph")
            for _ in range(5):
                out = net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 30
            for _ in range(n):
                out = net(x)
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / n * 1e3
            res[mode + "_out"] = [t.clone() for t in out]
        same = all(torch.equal(a, c) for a, c in zip(res["eager_out"], res["graph_out"]))
        print("size %d batch %d: eager %.3f ms  graph %.3f ms  (%.1f / %.1f fps)  identical=%s" %
              (size, b, res["eager"], res["graph"], b / res["eager"] * 1e3, b / res["graph"] * 1e3, same))
