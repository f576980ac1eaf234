"""The read-only pass of component 0 on the caller's column-major X (NIPALS, A = 1: the fit's only fused launch) against the grid size."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, pls_amd
N, K = 1 << 20, 512
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 1); h.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_NIPALS)
X = h.synth_x(0, N, K, 1); Y = h.synth_y(0, N, 1, 1)
out = None
for grid in (0, 256, 512, 768, 1024, 2048, 0):
    h.set_option(pls_amd.OPT_FUSED_GRID, grid)
    out = h.fit_device(X, Y, 1, out=out); torch.cuda.synchronize(); h.timing()
    for _ in range(10): h.fit_device(X, Y, 1, out=out)
    tm = h.timing()
    print("grid=%d: fused %.4f ms over %d launches; xty %.4f ms" % (grid, tm['ms']['fused'] / max(tm['launches']['fused'], 1), tm['launches']['fused'], tm['ms']['xty'] / max(tm['launches']['xty'], 1)), flush=True)
