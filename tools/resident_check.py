#!/usr/bin/env python3
"""The one-launch resident fits of mid-size data against the oracle and against the general plan, with times: "gram" = the X^T X form
AUTO takes for K <= 128 (three grid-wide hand-offs in all, resident_gram.hpp), "resident" = one exchange per component (KERNEL plan).
python tools/resident_check.py [out.txt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pls_amd
from oracle import pls_oracle as po
one = po.OracleLib(omp=False)
h = pls_amd.Handle()
bad = 0
lines = []
for (N, K, A, dt, M) in ((1025, 26, 5, "f64", 1), (2000, 30, 5, "f64", 1), (5000, 128, 10, "f64", 1), (5000, 20, 5, "f64", 1), (8000, 100, 5, "f64", 1), (4000, 400, 5, "f64", 1),
                      (20000, 16, 5, "f64", 1), (10000, 64, 8, "f64", 1), (100000, 40, 12, "f64", 1), (200000, 26, 6, "f64", 1), (3001, 77, 7, "f32", 1), (65537, 50, 9, "f32", 1),
                      (1025, 26, 5, "f64", 3), (2000, 30, 5, "f64", 2), (5000, 128, 10, "f64", 4), (8000, 100, 5, "f64", 8), (4000, 400, 5, "f64", 3), (20000, 16, 5, "f64", 5),
                      (100000, 40, 12, "f64", 8), (200000, 26, 6, "f64", 2), (3001, 77, 7, "f32", 3), (65537, 50, 9, "f32", 8)):
    tdt = torch.float64 if dt == "f64" else torch.float32
    X = h.synth_x(3, N, K, 11, dtype=tdt); Y = h.synth_y(3, N, M, 11, dtype=tdt)
    Xh = X.cpu().numpy().astype(np.float64); Yh = Y.cpu().numpy().astype(np.float64)
    ref = one.plsr(Xh, Yh, A); Bref = one.coefficients(ref["R"], ref["Q"])
    res = {}
    for mode in ("g", "1", "0"):
        os.environ["PLS_HIP_RESIDENT"] = "0" if mode == "0" else "1"
        hh = pls_amd.Handle()
        if mode == "g": hh.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_AUTO)
        out = hh.fit_device(X, Y, A); hh.synchronize()
        us = 1e30
        for _ in range(5):  # (the best of five runs of 40: a stall of some tens of ms now and then is the script's, not the kernels')
            t0 = time.perf_counter()
            for _ in range(40): hh.fit_device(X, Y, A, out=out)
            hh.synchronize(); us = min(us, (time.perf_counter() - t0) / 40 * 1e6)
        B = out["B"].cpu().numpy(); T = out["T"].cpu().numpy().astype(np.float64)
        eb = po.rel_fro(B, Bref)
        et = max(po.rel_fro(np.sign(T[:, a] @ ref["T"][:, a]) * T[:, a], ref["T"][:, a]) for a in range(A))
        res[mode] = (us, eb, et)
        hh.close()
    tol = 1e-10 if dt == "f64" else 2e-5
    ttol = 1e-8 if dt == "f64" else 1e-3
    flag = "" if all(res[m][1] < tol and res[m][2] < ttol for m in ("g", "1")) else "   <-- BAD"
    bad += bool(flag)
    line = "N=%7d K=%4d M=%d A=%2d %s  AUTO %7.1f us (B err %.1e, T err %.1e)   resident %7.1f us (B err %.1e, T err %.1e)   general plan %7.1f us   x%.2f / x%.2f%s" % (
        N, K, M, A, dt, res["g"][0], res["g"][1], res["g"][2], res["1"][0], res["1"][1], res["1"][2], res["0"][0], res["0"][0] / res["g"][0], res["0"][0] / res["1"][0], flag)
    print(line, flush=True); lines.append(line)
print("done:", bad, "bad")
if len(sys.argv) > 1: open(sys.argv[1], "w").write("\n".join(lines) + "\n")
