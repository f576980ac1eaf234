#!/usr/bin/env python3
"""Where a SHORT deflating pass spends its time: wall-clock stamps of every workgroup (start, tile loop done, tail entered,
exit) of the last pass of a fit -- testing/libpls_hip.so only (PLS_HIP_TESTING).  python tools/pass_stamps.py [rows] [algo]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ["PLS_AMD_LIBRARY"] = os.path.join(ROOT, "pls_amd", "csrc", "testing", "libpls_hip.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch

import pls_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
algo = {"nipals": pls_amd.ALGO_NIPALS, "kernel": pls_amd.ALGO_KERNEL}[sys.argv[2] if len(sys.argv) > 2 else "nipals"]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
M = int(sys.argv[4]) if len(sys.argv) > 4 else 1
A = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dt = torch.float32 if (len(sys.argv) > 6 and sys.argv[6] == "f32") else torch.float64
h = pls_amd.Handle()
h.set_option(pls_amd.OPT_ALGO, algo)
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT, dtype=dt)
Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT, dtype=dt)
import ctypes

L = pls_amd.lib()
L.pls_hip_test_set_pass_stamps.argtypes = [ctypes.c_void_p]  # (without it ctypes would pass the pointer as a 32-bit int)
L.pls_hip_test_set_pass_stamps.restype = ctypes.c_int
buf = torch.zeros(8 * 2048, dtype=torch.int64, device="cuda")
for _ in range(3):
    h.fit_device(X, Y, A)
h.synchronize()
assert L.pls_hip_test_set_pass_stamps(buf.data_ptr()) == 0
h.fit_device(X, Y, A)   # every pass overwrites the stamps: what is left are those of the last pass
h.synchronize()
L.pls_hip_test_set_pass_stamps(None)
full = buf.cpu().numpy().reshape(-1, 8)
s = full[:, :4].astype(np.float64)
s = s[s[:, 0] > 0]
keep = s[:, 0] >= s[:, 0].max() - 5000
xcc = (full[:, 4][full[:, 0] > 0][keep] & 0xf)
hwid = full[:, 5][full[:, 0] > 0][keep]
s = s[keep]  # the LAST pass only (earlier, larger grids left their stamps in the rows behind): 50 us of starts
tick = 1e6 / 100e6  # wall_clock64: 100 MHz -> us
t0 = s[:, 0].min()
s = (s - t0) * tick
q = lambda x: "min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f" % (x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max())
print(f"{N} x {K} {str(dt)[6:]}, M = {M}, A = {A}, {sys.argv[2] if len(sys.argv) > 2 else 'nipals'} plan, last pass of a fit, {len(s)} workgroups; microseconds from the first workgroup's start")
print("start            ", q(s[:, 0]))
print("tile loop done   ", q(s[:, 1]))
print("loop duration    ", q(s[:, 1] - s[:, 0]))
if (s[:, 2] > 0).all():
    print("tail entered     ", q(s[:, 2]))
print("exit             ", q(s[:, 3]))
e = np.sort(s[:, 1])
print("workgroups still in their tile loop, by time before the last one ends:", ", ".join(
    "%d us: %d" % (d, int((e > e[-1] - d).sum())) for d in (1, 2, 4, 6, 8, 12, 16, 24)))
print("last exit - last loop end: %.2f us;  last loop end - median loop end: %.2f us" % (s[:, 3].max() - s[:, 1].max(), s[:, 1].max() - np.median(s[:, 1])))
d = s[:, 1] - s[:, 0]
print("loop duration by blockIdx % 8 (workgroups b and b + 8 share an XCD):", " ".join("%.1f" % d[i::8].mean() for i in range(8)),
      " spread inside a class: " + " ".join("%.1f" % d[i::8].std() for i in range(8)))
np.save(os.path.join(ROOT, "gpurun_out", "r5", "pass_stamps_%d_%s.npy" % (N, sys.argv[2] if len(sys.argv) > 2 else "nipals")), s)
print("XCC_ID of blockIdx 0..15:", " ".join(str(int(x)) for x in xcc[:16]), "; blockIdx % 8 == const per XCC:", all(len(set((np.arange(len(xcc))[xcc == x] % 8).tolist())) == 1 for x in set(xcc.tolist())))
print("loop duration by physical XCC_ID:", " ".join("%d: %.1f" % (x, d[xcc == x].mean()) for x in sorted(set(xcc.tolist()))))
