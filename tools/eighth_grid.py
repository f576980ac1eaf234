"""One rank's share of config 3 on 8 GPUs (131,072 x 512): fit time against the workgroup count of the fused pass
(OPT_FUSED_GRID; 0 = the library's choice), profile off (no event brackets), 30 fits each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
N, K, M, A = 131072, 512, 1, 20
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_ALGO, 1)
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
out = h.fit_device(X, Y, A); torch.cuda.synchronize()
for grid in (0, 128, 256, 384, 512, 768, 1024):
    h.set_option(pls_amd.OPT_FUSED_GRID, grid)
    for _ in range(5): h.fit_device(X, Y, A, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): h.fit_device(X, Y, A, out=out)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 30 * 1e3
    print(f"grid={grid or 'auto':>5}: {ms:.3f} ms per fit = {ms/A*1e3:.1f} us per component, {A/ms*1e3:.0f} components/s", flush=True)
