"""CPU restatement of the OPT-IN split-fp32 conv arithmetic of videoyolo_amd/csrc/conv_split.hip — TEST INFRASTRUCTURE,
like everything under oracle/: only tests/ may import it; the product never does.

This is not a restatement of the reference (the reference has no such mode: mxnet hands `Conv2D`,
models/definitions/layers.py:63-70, to cuDNN / MKL-DNN); it pins what the kernel claims to compute, independently of
the kernel: every fp32 operand is cut into three bf16 numbers by round-to-nearest-even,

    h = bf16(x),   m = bf16(x - h),   l = bf16(x - h - m)                 (both differences are exact in fp32)

and a product x * w is taken as the six partial products  l h + h l + m m + m h + h m + h h  (the three left out are
below 2^-25 |x w|).  `conv_split_ref` evaluates exactly those six products in float64 — so the only thing the GPU result
may differ by is the rounding of its fp32 accumulation; `conv_split_ref(..., products=FIVE)` drops `h_x l_w`, which is how
the tests show that they would notice a missing product.  Parity status of this file: checked against float64 and against
its own exactness properties (tests/test_split_oracle.py); there is nothing in the reference to pin it to.
"""
import numpy as np

SIX = ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0))   # (plane of x, plane of w); plane 0 = h, 1 = m, 2 = l
FIVE = tuple(p for p in SIX if p != (0, 2))


def bf16_rne(x):
    """fp32 -> the nearest bf16 value (ties to even), returned as fp32.  Same integer formula as
    split_weights_kernel; finite inputs only."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32).reshape(np.shape(x))


def split3(x):
    """x (fp32) -> (h, m, l), each holding bf16 values, with h + m + l == x exactly."""
    x = np.ascontiguousarray(x, np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)      # exact: h is x rounded to 8 significant bits
    m = bf16_rne(r)
    l = bf16_rne((r - m).astype(np.float32))
    return h, m, l


def conv_split_ref(x, w, stride, pad, products=SIX):
    """The sum of the selected partial products of conv2d(x, w) in float64.  x (B,Cin,H,W), w (Cout,Cin,k,k) fp32."""
    import torch
    import torch.nn.functional as F
    xs = [torch.from_numpy(np.ascontiguousarray(p, np.float64)) for p in split3(x)]
    ws = [torch.from_numpy(np.ascontiguousarray(p, np.float64)) for p in split3(w)]
    out = None
    with torch.no_grad():
        for px, pw in products:
            t = F.conv2d(xs[px], ws[pw], None, stride, pad)
            out = t if out is None else out + t
    return out.numpy()


def abs_product_sum(x, w, stride, pad):
    """sum_k |x_k w_k| per output (float64): the scale the fp32 accumulation error is relative to."""
    import torch
    import torch.nn.functional as F
    with torch.no_grad():
        return F.conv2d(torch.from_numpy(np.abs(x).astype(np.float64)), torch.from_numpy(np.abs(w).astype(np.float64)),
                        None, stride, pad).numpy()
