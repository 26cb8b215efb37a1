"""Stream-K launches against plain launches over a sweep of (size, batch): two processes per point (VY_CONV_SK=1 / 0), the
SHA-256 of heads + detections (and, for sizes that are multiples of 32 with --train, of losses + gradients) must be equal.
usage: python tools/sk_sweep.py [--train]      (on a GPU box; prints one line per point, exits 1 on the first mismatch)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "sk_digest_worker.py")
train = "--train" in sys.argv
points = [(320, 24), (352, 16), (416, 5), (416, 32), (480, 8), (512, 12), (544, 3), (608, 2), (608, 4), (608, 16),
          (608, 24), (640, 7), (704, 6), (800, 4), (1024, 2)]
bad = 0
for size, batch in points:
    got = {}
    for sk in ("1", "0"):
        args = [sys.executable, WORKER, str(size), str(batch)] + ([] if train and size % 32 == 0 else ["infer"])
        p = subprocess.run(args, env=dict(os.environ, VY_CONV_SK=sk), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=900, universal_newlines=True)
        if p.returncode != 0:
            print(p.stdout[-2000:])
            sys.exit(2)
        got[sk] = json.loads([l for l in p.stdout.splitlines() if l.startswith("DIGEST ")][-1][7:])
    ok = got["1"]["infer"] == got["0"]["infer"] and got["1"].get("train") == got["0"].get("train") and not got["0"]["sk_launches"]
    print("%4d x %-3d  stream-K launches %2d of %d  %s%s" % (size, batch, len(got["1"]["sk_launches"]), got["1"]["conv_launches"],
                                                         "equal" if ok else "DIFFERENT", "  (+ training step)" if "train" in got["1"] else ""))
    sys.stdout.flush()
    bad += not ok
sys.exit(1 if bad else 0)
