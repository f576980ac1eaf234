"""Repeated fits of odd-sized matrices against the oracle (regression check of the tail-rows / padded-sweep paths).
    python tools/stress_odd.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pls_amd
from oracle import pls_oracle as po
ora = po.OracleLib()
h = pls_amd.Handle()
bad = 0
cases = [(1365, 1024, 1, 20), (4097, 1024, 4, 12), (1365, 1024, 4, 5), (1365, 1024, 4, 9), (1366, 1024, 4, 20), (1365, 1024, 4, 20),
         (1365, 520, 4, 20), (1365, 500, 4, 20), (1365, 1100, 4, 20), (1381, 1024, 4, 20), (1365, 128, 2, 10), (1365, 200, 2, 10)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (N, K, M, A) in cases:
    Xh, Yh = ora.synth_x(0, N, K), ora.synth_y(0, N, M)
    ref = ora.plsr(Xh, Yh, A); Bref = ora.coefficients(ref["R"], ref["Q"])
    alt = ora.plsr(Xh, Yh, A, nipals=True); cond = po.rel_fro(ora.coefficients(alt["R"], alt["Q"]), Bref)
    X = pls_amd.as_colmajor(torch.from_numpy(Xh).cuda()); Y = pls_amd.as_colmajor(torch.from_numpy(Yh).cuda())
    for algo in (1, 0):
        h.set_option(pls_amd.OPT_ALGO, algo)
        errs = []
        out = None
        for rep in range(5):
            out = h.fit_device(X, Y, A, out=out); h.synchronize()
            errs.append(po.rel_fro(out["B"].cpu().numpy(), Bref))
        nb = sum(e > max(1e-10, 50 * cond) for e in errs)
        bad += nb
        print(N, K, M, A, "algo", algo, "max err %.3e" % max(errs), "cpu-route spread %.1e" % cond, "bad", nb, flush=True)
print("total bad", bad)
sys.exit(1 if bad else 0)
