"""Chain-latency floor of a small-batch frame.  Under the pinned summation order every conv output is ONE fma chain of
K = k*k*Cin terms, so a layer cannot finish before K dependent multiply-adds have retired, however many CUs are idle:
v_mfma_f32_32x32x2_f32 retires 2 terms per 64 cycles (32 cycles per term), v_mfma_f32_16x16x4_f32 4 terms per 40
dependent cycles (10 per term).  usage: python tools/chain_floor.py gpurun_out/r03b_lay_b1_off.txt  (a layer_profile table)"""
import sys

CLK = 2.4e9
rows = [l.split() for l in open(sys.argv[1]) if "|" in l]
t32 = t16 = meas = 0.0
for r in rows:
    o, i, k, _ = map(int, r[1].split("x"))
    K = i * k * k
    t32 += K * 32 / CLK
    t16 += K * 10 / CLK
    meas += float(r[2])
print("%d conv launches: measured (sum of launches, HIP events) %.3f ms" % (len(rows), meas))
print("chain floor, v_mfma_f32_32x32x2_f32 (32 cycles per term): %.3f ms = %.1f TFLOP/s at 139.76 GFLOP" % (t32 * 1e3, 139.76 / t32 / 1e3))
print("chain floor, v_mfma_f32_16x16x4_f32 (10 cycles per term): %.3f ms" % (t16 * 1e3))
