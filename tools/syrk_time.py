#!/usr/bin/env python3
"""The SYRK launch (X^T X on the fp64 matrix cores, X^T Y on board) by HIP events over many launches: KERNEL_TYPE2 fits of ONE
component on config 3 -- the "xty" event family of such a fit is the SYRK alone.  python tools/syrk_time.py [launches] [f32] [K] [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
M = int(sys.argv[4]) if len(sys.argv) > 4 else 1
N = (1 << 29) // K
h = pls_amd.Handle()
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT, dtype=dt); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT, dtype=dt)
out = None
for _ in range(5):
    out = h.fit_device(X, Y, 1, method=pls_amd.KERNEL_TYPE2, out=out)
h.synchronize()
h.set_option(pls_amd.OPT_PROFILE, 1)
h.timing()
for _ in range(n):
    out = h.fit_device(X, Y, 1, method=pls_amd.KERNEL_TYPE2, out=out)
h.synchronize()
t = h.timing()
ms = t["ms"]["xty"] / max(t["launches"]["xty"], 1)
nbk = (K + 127) // 128
tiles = 64 * (nbk * (nbk - 1) // 2) + 36 * nbk
gf = 2.0 * N * 256 * tiles / 1e9
print("SYRK %d x %d %s M=%d: %.4f ms over %d launches -> %.1f GF executed, %.2f TF = %.3f of 78.6" % (N, K, str(dt)[6:], M, ms, t["launches"]["xty"], gf, gf / ms, gf / ms / 78.6))
