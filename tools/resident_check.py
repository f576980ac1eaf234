#!/usr/bin/env python3
"""The one-launch resident fit (mid-size single-response data) against the oracle and against the general plan, with times.
python tools/resident_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pls_amd
from oracle import pls_oracle as po
one = po.OracleLib(omp=False)
h = pls_amd.Handle()
bad = 0
for (N, K, A, dt, M) in ((1025, 26, 5, "f64", 1), (2000, 30, 5, "f64", 1), (5000, 128, 10, "f64", 1), (5000, 20, 5, "f64", 1), (8000, 100, 5, "f64", 1), (4000, 400, 5, "f64", 1),
                      (20000, 16, 5, "f64", 1), (10000, 64, 8, "f64", 1), (100000, 40, 12, "f64", 1), (200000, 26, 6, "f64", 1), (3001, 77, 7, "f32", 1), (65537, 50, 9, "f32", 1),
                      (1025, 26, 5, "f64", 3), (2000, 30, 5, "f64", 2), (5000, 128, 10, "f64", 4), (8000, 100, 5, "f64", 8), (4000, 400, 5, "f64", 3), (20000, 16, 5, "f64", 5),
                      (100000, 40, 12, "f64", 8), (200000, 26, 6, "f64", 2), (3001, 77, 7, "f32", 3), (65537, 50, 9, "f32", 8)):
    tdt = torch.float64 if dt == "f64" else torch.float32
    X = h.synth_x(3, N, K, 11, dtype=tdt); Y = h.synth_y(3, N, M, 11, dtype=tdt)
    Xh = X.cpu().numpy().astype(np.float64); Yh = Y.cpu().numpy().astype(np.float64)
    ref = one.plsr(Xh, Yh, A); Bref = one.coefficients(ref["R"], ref["Q"])
    res = {}
    for mode in ("1", "0"):
        os.environ["PLS_HIP_RESIDENT"] = mode
        hh = pls_amd.Handle()
        out = hh.fit_device(X, Y, A); hh.synchronize()
        reps = 100
        t0 = time.perf_counter()
        for _ in range(reps): hh.fit_device(X, Y, A, out=out)
        hh.synchronize(); us = (time.perf_counter() - t0) / reps * 1e6
        B = out["B"].cpu().numpy(); T = out["T"].cpu().numpy().astype(np.float64)
        eb = po.rel_fro(B, Bref)
        et = max(po.rel_fro(np.sign(T[:, a] @ ref["T"][:, a]) * T[:, a], ref["T"][:, a]) for a in range(A))
        res[mode] = (us, eb, et)
        hh.close()
    tol = 1e-10 if dt == "f64" else 2e-5
    flag = "" if res["1"][1] < tol and res["1"][2] < (1e-8 if dt == "f64" else 1e-3) else "   <-- BAD"
    bad += bool(flag)
    print("N=%7d K=%4d M=%d A=%2d %s  resident %8.1f us (B err %.1e, T err %.1e)   general plan %8.1f us (B err %.1e)   x%.2f%s" % (
        N, K, M, A, dt, res["1"][0], res["1"][1], res["1"][2], res["0"][0], res["0"][1], res["0"][0] / res["1"][0], flag), flush=True)
print("done:", bad, "bad")
