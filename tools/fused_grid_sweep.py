"""Fused-pass launch time against the grid size (PLS_HIP_OPT_FUSED_GRID), NIPALS plan (read+write passes on the
row-tile-major working copy + one read-only pass per fit) and KERNEL plan (read-only passes on the caller's X).
usage: fused_grid_sweep.py [N K f64|f32]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dt = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else torch.float64
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 1)
A = 8
X = h.synth_x(0, N, K, 1, dtype=dt); Y = h.synth_y(0, N, 1, 1, dtype=dt)
for algo, name in ((1, "nipals"), (0, "kernel")):
    h.set_option(pls_amd.OPT_ALGO, algo)
    out = None
    for grid in (0, 192, 256, 384, 512, 1024, 256, 512):
        h.set_option(pls_amd.OPT_FUSED_GRID, grid)
        out = h.fit_device(X, Y, A, out=out); torch.cuda.synchronize(); h.timing()
        for _ in range(3): h.fit_device(X, Y, A, out=out)
        tm = h.timing()
        us = 1e3 * tm['ms']['fused'] / tm['launches']['fused']
        gb = tm['bytes']['fused'] / tm['launches']['fused'] / 1e9
        print(f"{name} N={N} K={K} {str(dt)[6:]} grid={grid}: fused launch avg {us:8.1f} us  {gb/us*1e3:.2f} TB/s", flush=True)
