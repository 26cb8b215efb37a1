"""fp32 error of the 1-D Winograd forms along x — F(2, 3) (what conv_wino.hip computes: 4 products per 2 outputs, i.e. 2/3 of
the multiplications) and F(4, 3) (6 per 4 outputs: 1/2) — against a float64 direct 3x3 conv, beside the direct fp32 conv.
numpy, random-normal data, He-scaled weights.   usage: python tools/numerics/winograd_1d_error.py"""
import numpy as np

rng = np.random.default_rng(0)
F23 = (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64),
       np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64),
       np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64))
F43 = (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                 [0, 4, 0, -5, 0, 1]], np.float64),
       np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                 [0, 0, 1]], np.float64),
       np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64))


def run(H, W, Cin, Cout):
    x = np.zeros((H + 2, W + 2 + 4, Cin), np.float32)
    x[1:H + 1, 1:W + 1] = rng.standard_normal((H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, 3, 3, Cin)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)

    def direct(dt):
        out = np.zeros((H, W, Cout), dt)
        for dy in range(3):
            for dx in range(3):
                out += x[dy:dy + H, dx:dx + W, :].astype(dt) @ w[:, dy, dx, :].astype(dt).T
        return out
    ref, d32 = direct(np.float64), direct(np.float32)
    res = {"direct fp32": d32}
    for name, (BT, G, AT), m in (("F(2,3)", F23, 2), ("F(4,3)", F43, 4)):
        a = m + 2
        BT32, G32, AT32 = BT.astype(np.float32), G.astype(np.float32), AT.astype(np.float32)
        U = np.einsum('ik,odkc->odic', G32, w).astype(np.float32)           # (Cout, dy, xi, Cin)
        out = np.zeros((H, W + m, Cout), np.float32)
        for x0 in range(0, W, m):
            d = x[:, x0:x0 + a, :]                                          # rows, a pixels, Cin
            V = np.einsum('ij,rjc->ric', BT32, d).astype(np.float32)        # rows, xi, Cin
            M = np.zeros((H, a, Cout), np.float32)
            for dy in range(3):
                M += np.einsum('ric,oic->rio', V[dy:dy + H], U[:, dy]).astype(np.float32)
            out[:, x0:x0 + m, :] = np.einsum('ji,rio->rjo', AT32, M).astype(np.float32)
        res[name] = out[:, :W]
    line = "H=%d W=%d Cin=%d Cout=%d K=%d | max |value| %.2f" % (H, W, Cin, Cout, 9 * Cin, np.abs(ref).max())
    for k, v in res.items():
        line += " | %s: max %.2e rms %.2e" % (k, np.abs(v - ref).max(), np.sqrt(((v - ref) ** 2).mean()))
    print(line)


run(24, 24, 64, 64)
run(24, 24, 128, 64)
run(12, 12, 512, 64)
