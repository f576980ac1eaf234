#!/usr/bin/env python3
"""The scores T = X R of config 3 (src/pls.cpp:439-442; 1,048,576 x 512 times 512 x 20, the GRAM plan's last sweep) by HIP
events over many launches of pls_hip_xb.  python tools/xb_time.py [launches] [f32] [K] [columns]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
C = int(sys.argv[4]) if len(sys.argv) > 4 else 20
N = (1 << 29) // K
es = 4 if dt == torch.float32 else 8
h = pls_amd.Handle()
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT, dtype=dt)
B = pls_amd.as_colmajor(torch.randn(K, C, dtype=torch.float64, device="cuda"))
for _ in range(5): out = h.xb(X, B)
h.synchronize()
h.set_option(pls_amd.OPT_PROFILE, 1)
h.timing()
for _ in range(n): out = h.xb(X, B)
h.synchronize()
t = h.timing()
ms = t["ms"]["xb"] / max(t["launches"]["xb"], 1)
by = N * K * es + N * C * es + K * C * 8
print("X B %d x %d %s, %d columns: %.4f ms over %d launches -> %.2f TB/s = %.3f of 8 TB/s (X once + the output)" % (
    N, K, str(dt)[6:], C, ms, t["launches"]["xb"], by / ms / 1e9, by / ms / 8e9))
