#!/usr/bin/env python3
"""retile_xty_kernel (the KERNEL plan's first sweep: copy into tiles + X^T Y) against the placement of the copy relative to X.
One process per offset (PLS_HIP_EXP_WORK_OFF is read once): python tools/retile_offset.py <offset bytes> [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
N, K, M, A, dt = {"C3": (1 << 20, 512, 1, 20, torch.float64), "C4": (131072, 4096, 8, 50, torch.float32)}[sys.argv[2] if len(sys.argv) > 2 else "C3"]
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_KERNEL)
pad = int(os.environ.get("XPAD", "0"))
if pad: _hold = torch.empty(pad, dtype=torch.uint8, device="cuda")  # shifts where the caller's X lands
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT, dtype=dt); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT, dtype=dt)
out = h.fit_device(X, Y, A); h.synchronize()
h.set_option(pls_amd.OPT_PROFILE, 1); h.timing()
for _ in range(6): out = h.fit_device(X, Y, A, out=out)
h.synchronize(); t = h.timing()
print("off %10d  X %x  retile_xty %.4f ms  pass %.4f ms  fit %.3f ms" % (int(os.environ.get("PLS_HIP_EXP_WORK_OFF", "0")), X.data_ptr(),
      t["ms"]["deflate"] / max(t["launches"]["deflate"], 1), t["ms"]["fused"] / max(t["launches"]["fused"], 1), t["fit_ms"] / max(t["fits"], 1)))
