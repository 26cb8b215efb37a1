"""Builds videoyolo_amd/libvyolo.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m videoyolo_amd.build          # incremental
    python -m videoyolo_amd.build --force  # rebuild everything

Flags that matter for parity (include/vy_math.h): -ffp-contract=off (no implicit fma),
correctly rounded fp32 divide/sqrt, fp32 denormals kept (hipcc default).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libvyolo.so")
SOURCES = ["net.hip", "train.hip", "conv_igemm.hip", "conv_split.hip", "conv_wino.hip", "conv_small.hip", "wgrad.hip", "wgrad_split.hip", "misc_kernels.hip", "train_kernels.hip", "detect.hip", "preproc.hip", "targets.hip"]
HEADERS = [os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "net_internal.h"), os.path.join(CSRC, "conv_device.h"), os.path.join(CSRC, "sk_schedule.h"), os.path.join(CSRC, "conv_cost_model.h"), os.path.join(CSRC, "split_device.h"),
           os.path.join(HERE, "..", "include", "vyolo.h"),
           os.path.join(HERE, "..", "include", "vy_math.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]
# measurement builds only (tools/ab_bn_bounds.sh: VY_BUILD_EXTRA_FLAGS=-DVY_TRAIN_ABL_BUILD); use with --force
FLAGS += os.environ.get("VY_BUILD_EXTRA_FLAGS", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + HEADERS):
            cmd = [HIPCC] + FLAGS + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (src, out.decode(errors="replace")))
        elif verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("libvyolo.so build failed")
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
