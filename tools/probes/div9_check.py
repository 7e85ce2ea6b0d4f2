"""Exhaustive check that q = fma(r, y, q0), q0 = a*y, r = fma(-b, q0, a), y = fp32(1/b) equals the correctly
rounded fp32 quotient a/b for every mantissa (b = 9: the SSIM 3x3 mean of cost_volume.hip)."""
import numpy as np
f32 = np.float32
for b in (9,):
    y = f32(1.0) / f32(b)
    lo, hi = np.array([2.0 ** -20], f32).view(np.uint32)[0], np.array([32.0], f32).view(np.uint32)[0]
    bad = 0
    for s in range(int(lo), int(hi), 1 << 24):
        a = np.arange(s, min(s + (1 << 24), int(hi)), dtype=np.uint32).view(f32)
        want = (a / f32(b)).astype(f32)
        q0 = (a * y).astype(f32)
        r = (a.astype(np.float64) - np.float64(b) * q0.astype(np.float64)).astype(f32)
        q1 = (q0.astype(np.float64) + r.astype(np.float64) * np.float64(y)).astype(f32)
        bad += int((q1 != want).sum())
    print(f"b={b}: y={float(y).hex()} mismatches={bad}")
