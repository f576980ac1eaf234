"""Many responses: one fit on the GPU (default plan) against the CPU restatement (all cores of the job's share), per shape.
usage: many_responses_scan.py [out.txt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pls_amd
from oracle import pls_oracle as po
one = po.OracleLib(omp=True)
h = pls_amd.Handle()
lines = []
for (N, K) in ((20000, 256), (2000, 2000), (200000, 64), (500, 6000)):
    for M in (16, 40, 49, 64, 100, 300, 600):
        A = 8
        X = one.synth_x(3, N, K); Y = one.synth_y(3, N, M)
        Xd = pls_amd.as_colmajor(torch.from_numpy(X).cuda()); Yd = pls_amd.as_colmajor(torch.from_numpy(Y).cuda())
        out = h.fit_device(Xd, Yd, A); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): h.fit_device(Xd, Yd, A, out=out)
        torch.cuda.synchronize(); g = (time.perf_counter() - t0) / 5 * 1e3
        t0 = time.perf_counter(); ref = one.plsr(X, Y, A); c = (time.perf_counter() - t0) * 1e3
        err = po.rel_fro(out["B"].cpu().numpy(), one.coefficients(ref["R"], ref["Q"]))
        line = "N=%7d K=%5d M=%5d A=%d   GPU %9.3f ms   CPU (OpenMP) %10.3f ms   x%7.1f   B rel err %.1e" % (N, K, M, A, g, c, c / g, err)
        print(line, flush=True); lines.append(line)
if len(sys.argv) > 1: open(sys.argv[1], "w").write("\n".join(lines) + "\n")
