"""The north star's named step on its own: X -= t p^T (pls_hip_deflate -> deflate_piece_kernel) on the caller's
column-major matrices at config 3's shape, 20 launches in place -- the command the rocprofv3 kernel-trace and PMC
passes of profiles/rNN/deflate_* run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
N, K = 1 << 20, 512
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 1)
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT); Y = h.synth_y(0, N, 1, pls_amd.SEED_DEFAULT)
out = h.fit_device(X, Y, 1); torch.cuda.synchronize()
t = out["T"][:, 0].contiguous(); p = out["P"][:, 0].contiguous()
W = pls_amd.colmajor_empty(N, K, torch.float64, X.device); W.copy_(X)
for _ in range(3): h.deflate(W, t, p, dst=W)
torch.cuda.synchronize(); h.timing()
for _ in range(20): h.deflate(W, t, p, dst=W)
tm = h.timing()
ms = tm["ms"]["deflate"] / tm["launches"]["deflate"]; gb = tm["bytes"]["deflate"] / tm["launches"]["deflate"] / 1e9
print(f"deflate_piece_kernel, in place, N={N} K={K} fp64: {ms:.4f} ms per launch (HIP events), {gb:.3f} GB algorithmic, "
      f"{gb/ms*1e3:.0f} GB/s = {gb/ms/8:.3f} of the 8 TB/s peak", flush=True)
