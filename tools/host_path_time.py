"""Time the host-memory entry of the fit (what PLS::Model(const Mat2D&, ...) pays): H2D of X, Y, the
fit, D2H of the results -- BASELINE config 3.  Reported in BASELINE.md / DESIGN.md, never as bench `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pls_amd
N, K, M, A = 1 << 20, 512, 1, 20
h = pls_amd.Handle()
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
Xh = np.asfortranarray(X.cpu().numpy()); Yh = np.asfortranarray(Y.cpu().numpy())
del X, Y
for algo in (pls_amd.ALGO_KERNEL, pls_amd.ALGO_NIPALS, pls_amd.ALGO_AUTO):
    h.set_option(pls_amd.OPT_ALGO, algo)
    h.fit_host(Xh, Yh, A)
    t0 = time.perf_counter(); out = h.fit_host(Xh, Yh, A); dt = time.perf_counter() - t0
    print(f"algo={algo} host-memory fit incl. H2D/D2H: {dt*1e3:.1f} ms -> {A/dt:.1f} components/s "
          f"(X = {Xh.nbytes/1e9:.2f} GB pageable -> {Xh.nbytes/dt/1e9:.1f} GB/s effective)")
