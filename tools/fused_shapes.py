"""Fused-pass launch time (HIP events, profile level 1) over storage types and widths: NIPALS (read+write) and
KERNEL (read-only) plans."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 1)
for algo in (1, 0):
    h.set_option(pls_amd.OPT_ALGO, algo)
    for (N, K, dt) in ((1 << 20, 512, torch.float32), (1 << 20, 1024, torch.float32), (1 << 20, 768, torch.float32), (1 << 20, 128, torch.float32), (1 << 20, 256, torch.float64), (1 << 20, 512, torch.float64), (1 << 19, 1024, torch.float64), (1 << 21, 128, torch.float64)):
        X = h.synth_x(0, N, K, 1, dtype=dt); Y = h.synth_y(0, N, 1, 1, dtype=dt)
        out = h.fit_device(X, Y, 6); torch.cuda.synchronize(); h.timing()
        for _ in range(3): h.fit_device(X, Y, 6, out=out)
        tm = h.timing()
        us = 1e3 * tm['ms']['fused'] / max(tm['launches']['fused'], 1)
        gb = tm['bytes']['fused'] / max(tm['launches']['fused'], 1) / 1e9
        print(f"algo={algo} N={N} K={K} {str(dt)[6:]}: fused launch avg {us:8.1f} us  {gb/us*1e3:.2f} TB/s  ({tm['launches']['fused']} launches)", flush=True)
        del X, Y, out
