#!/usr/bin/env python3
"""Per-workgroup wall-clock stamps of the SYRK launch (testing build): duration by kind of workgroup and by XCD.
python tools/syrk_stamps.py [K] [M]"""
import os, sys, ctypes
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ["PLS_AMD_LIBRARY"] = os.path.join(ROOT, "pls_amd", "csrc", "testing", "libpls_hip.so")
sys.path.insert(0, ROOT)
import numpy as np, torch, pls_amd
K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = (1 << 29) // K
h = pls_amd.Handle()
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
L = pls_amd.lib(); L.pls_hip_test_set_pass_stamps.argtypes = [ctypes.c_void_p]; L.pls_hip_test_set_pass_stamps.restype = ctypes.c_int
buf = torch.zeros(8 * 2048, dtype=torch.int64, device="cuda")
out = None
for _ in range(3): out = h.fit_device(X, Y, 1, method=pls_amd.KERNEL_TYPE2, out=out)
h.synchronize()
assert L.pls_hip_test_set_pass_stamps(buf.data_ptr()) == 0
out = h.fit_device(X, Y, 1, method=pls_amd.KERNEL_TYPE2, out=out); h.synchronize()
L.pls_hip_test_set_pass_stamps(None)
s = buf.cpu().numpy().reshape(-1, 8)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
st = (s[:, 0] - t0) / 100.0; en = (s[:, 1] - t0) / 100.0; d = en - st
xcc = s[:, 4] & 0xf
n = len(s)
nbk = (K + 127) // 128; noff = nbk * (nbk - 1) // 2
print(f"SYRK {N} x {K}, M = {M}: {n} workgroups; launch {en.max():.1f} us; start max {st.max():.1f} us")
print("end time   min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
print("duration by XCC_ID:", " ".join("%d: %.0f" % (x, d[xcc == x].mean()) for x in sorted(set(xcc.tolist()))))
ids = np.arange(n)
# kinds: the first noff * so ids are off-diagonal (id % noff = block); so is unknown here: infer from the duration pattern is not needed -- print by id range deciles
for lo in range(0, n, max(1, n // 8)):
    hi = min(n, lo + max(1, n // 8))
    print("ids %4d-%4d: mean duration %.0f us, mean end %.0f us" % (lo, hi - 1, d[lo:hi].mean(), en[lo:hi].mean()))
np.save(os.path.join(ROOT, "gpurun_out", "r5", f"syrk_stamps_{K}.npy"), s)
