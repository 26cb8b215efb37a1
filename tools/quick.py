import sys, time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R+'/tests')
import numpy as np, torch
import videoyolo_amd as vy
from videoyolo_amd import init
from oracle import yolo3_oracle as O
from conftest import frames, VOC_CLASSES
params = init.synthetic_params(O.param_shapes(20), seed=233)
net = vy.yolo3_darknet53(VOC_CLASSES, pretrained_base=False); net.set_parameters(params); net.collect_params().reset_ctx('cuda:0')
x = frames(2, 96)
out = net(x, return_index=True); torch.cuda.synchronize()
orc = O.OracleYolo3(20, params)
ref = orc.raw_heads(x)
for i in range(3):
    got = net.read_head(i).cpu().numpy()
    print('head', i, got.shape, 'maxdiff', np.abs(got-ref[i]).max(), 'exact', np.array_equal(got, ref[i]))
for name in ["stages.0.0", "stages.0.1", "stages.0.2.body.0", "stages.0.2.body.1", "stages.0.14.body.1"]:
    a = net.read_activation(name).cpu().numpy(); print(name, a.shape, float(np.abs(a).mean()))
r = orc(x)
ids, sc, bb, keep = [t.cpu().numpy() for t in out]
print('keep equal', np.array_equal(keep, r[3]), 'ids equal', np.array_equal(ids, r[0]), 'scores maxdiff', np.abs(sc-r[1]).max(), 'bbox maxdiff', np.nanmax(np.abs(bb-r[2])))
print(keep[0,:10], r[3][0,:10])
