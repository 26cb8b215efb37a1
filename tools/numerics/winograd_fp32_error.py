"""Is Winograd F(2x2, 3x3) in fp32 accurate enough to be a candidate arithmetic for the 3x3 convs?  numpy, one layer shape at a
time: direct fp32 conv and Winograd fp32 conv against a float64 direct conv.
usage: python tools/numerics/winograd_fp32_error.py"""
import numpy as np
rng = np.random.default_rng(0)
def run(H, Cin, Cout):
    x = rng.standard_normal((H + 2, H + 2, Cin)).astype(np.float32); x[0] = x[-1] = 0; x[:, 0] = x[:, -1] = 0
    w = (rng.standard_normal((Cout, 3, 3, Cin)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    # float64 direct
    def direct(xx, ww, dt):
        out = np.zeros((H, H, Cout), dt)
        for dy in range(3):
            for dx in range(3):
                out += xx[dy:dy + H, dx:dx + H, :].astype(dt) @ ww[:, dy, dx, :].astype(dt).T
        return out
    ref = direct(x, w, np.float64)
    d32 = direct(x, w, np.float32)
    BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
    AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
    U = np.einsum('ij,ojkc,lk->oilc', G, w, G).astype(np.float32)          # (Cout,4,4,Cin)
    T = H // 2
    out = np.zeros((H, H, Cout), np.float32)
    for ty in range(T):
        for tx in range(T):
            d = x[2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4, :]                      # (4,4,Cin)
            V = np.einsum('ij,jkc,lk->ilc', BT, d, BT).astype(np.float32)    # (4,4,Cin)
            M = np.einsum('oilc,ilc->oil', U, V).astype(np.float32)          # (Cout,4,4)
            Y = np.einsum('ij,ojk,lk->oil', AT, M, AT).astype(np.float32)    # (Cout,2,2)
            out[2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2, :] = Y.transpose(1, 2, 0)
    e_d = np.abs(d32 - ref).max(); e_w = np.abs(out - ref).max()
    r_d = np.sqrt(((d32 - ref) ** 2).mean()); r_w = np.sqrt(((out - ref) ** 2).mean())
    print("H=%d Cin=%d Cout=%d K=%d | max |value| %.2f | direct fp32: max %.2e rms %.2e | winograd F(2,3) fp32: max %.2e rms %.2e | x%.1f" % (H, Cin, Cout, 9 * Cin, np.abs(ref).max(), e_d, r_d, e_w, r_w, r_w / r_d))
run(26, 64, 64)
run(26, 128, 64)
run(12, 512, 64)
