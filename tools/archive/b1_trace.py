import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, videoyolo_amd as vy
dev = torch.device("cuda", 0)
net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False)
net.initialize(init="synthetic", seed=233)
net.collect_params().reset_ctx(dev)
net.set_nms(0.45, 400, 100)
x = torch.randn((1, 3, 608, 608), generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(20):
    net(x)
torch.cuda.synchronize()
