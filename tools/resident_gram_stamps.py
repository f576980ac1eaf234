#!/usr/bin/env python3
"""Where the one-launch X^T X fit (resident_gram.hpp) spends its time: wall-clock stamps of its phases, testing/libpls_hip.so only.
python tools/resident_gram_stamps.py N K A [f32|f64] [M]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ["PLS_AMD_LIBRARY"] = os.path.join(ROOT, "pls_amd", "csrc", "testing", "libpls_hip.so")
sys.path.insert(0, ROOT)
import ctypes, numpy as np, torch, pls_amd
N, K, A = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = torch.float32 if len(sys.argv) > 4 and sys.argv[4] == "f32" else torch.float64
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_AUTO)
M = int(sys.argv[5]) if len(sys.argv) > 5 else 1
X = h.synth_x(3, N, K, 11, dtype=dt); Y = h.synth_y(3, N, M, 11, dtype=dt)
L = pls_amd.lib()
L.pls_hip_test_set_pass_stamps.argtypes = [ctypes.c_void_p]; L.pls_hip_test_set_pass_stamps.restype = ctypes.c_int
buf = torch.zeros(8 * 2048, dtype=torch.int64, device="cuda")
out = None
for _ in range(5): out = h.fit_device(X, Y, A, out=out)
h.synchronize()
assert L.pls_hip_test_set_pass_stamps(buf.data_ptr()) == 0
h.fit_device(X, Y, A, out=out); h.synchronize()
L.pls_hip_test_set_pass_stamps(None)
s = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min(); s = (s - t0) / 100.0  # 100 MHz -> us
names = ["start", "parts stored", "hand-off 1 passed", "slices summed + hand-off 2", "XX in LDS (wg 0)", "components done (wg 0)", "hand-off 3 passed", "scores stored"]
print(f"{N} x {K}, M = {M}, A = {A}, {str(dt)[6:]}: {len(s)} workgroups; microseconds from the first workgroup's start")
for i, n in enumerate(names):
    col = s[:, i] if i not in (4, 5) else s[:1, i]
    print("  %-30s wg0 %7.2f   all: min %7.2f  median %7.2f  max %7.2f" % (n, s[0, i], col.min(), np.median(col), col.max()))
