"""Sanity check beyond 4096 columns (semi-fused NIPALS sweep, one-product KERNEL plan) against the oracle."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
from oracle import pls_oracle as po
orc = po.OracleLib(omp=True)
h = pls_amd.Handle()
for (N, K, M, A, dt) in ((8192, 6000, 2, 5, torch.float32), (4096, 5000, 1, 4, torch.float64)):
    X = h.synth_x(0, N, K, 3, dtype=dt); Y = h.synth_y(0, N, M, 3, dtype=dt)
    Xh = X.cpu().numpy().astype(np.float64); Yh = Y.cpu().numpy().astype(np.float64)
    ref = orc.plsr(Xh, Yh, A); Bref = orc.coefficients(ref["R"], ref["Q"])
    for algo in (1, 0):
        h.set_option(pls_amd.OPT_ALGO, algo)
        out = h.fit_device(X, Y, A); torch.cuda.synchronize()
        err = po.rel_fro(out["B"].cpu().numpy().astype(np.float64), Bref)
        print(f"N={N} K={K} M={M} A={A} {str(dt)[6:]} algo={algo}: B rel err {err:.2e}", flush=True)
