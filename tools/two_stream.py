"""Half-batches on two HIP streams vs one launch sequence (inference).  The side stream is created with
hipStreamCreateWithFlags directly so that it is certain to be its own hardware queue.
usage: python tools/two_stream.py [--size 608] [--batch 64] [--steps 5]"""
import argparse, ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import videoyolo_amd as vy

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=608)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False)
net.initialize(init="synthetic", seed=233)
net.collect_params().reset_ctx("cuda:0")
x = torch.randn((a.batch, 3, a.size, a.size), device="cuda:0")
path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
hip = ctypes.CDLL(path)
st = ctypes.c_void_p()
assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
side = torch.cuda.ExternalStream(st.value, device="cuda:0")

def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    return a.batch * a.steps / (time.perf_counter() - t)

ref = net.detect(x, return_index=True)
f1 = timeit(lambda: net.detect(x))
net.detect_two_streams(x)          # builds the twin
net._twin["stream"] = side
two = net.detect_two_streams(x, return_index=True)
f2 = timeit(lambda: net.detect_two_streams(x))
same = all(torch.equal(p, q) for p, q in zip(ref, two))
print("one stream %.1f fps   two streams %.1f fps   identical results: %s" % (f1, f2, same))
