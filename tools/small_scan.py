"""Small fits either side of the single-launch kernels' limits: time per fit back to back on the GPU -- the C-ABI's default plan (KERNEL)
and AUTO, what PLS::Model asks for -- against one CPU core (the oracle), 5 components.   usage: small_scan.py [out.txt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pls_amd
from oracle import pls_oracle as po
one = po.OracleLib(omp=False)
h = pls_amd.Handle()
ha = pls_amd.Handle(); ha.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_AUTO)
lines = []
for M in (1, 3):
    for (N, K) in ((10, 15), (20, 10), (40, 20), (64, 32), (65, 32), (64, 33), (200, 50), (512, 50), (1024, 26), (1025, 26), (1024, 40), (2000, 30), (5000, 20), (8000, 100), (4000, 500), (300, 400), (300, 500),
                   (60, 401), (60, 2000), (100, 5000), (3000, 100), (10000, 64), (20000, 16)):
        A = 5
        X = one.synth_x(3, N, K); Y = one.synth_y(3, N, M)
        Xd = pls_amd.as_colmajor(torch.from_numpy(X).cuda()); Yd = pls_amd.as_colmajor(torch.from_numpy(Y).cuda())
        def gpu_us(hh):
            out = hh.fit_device(Xd, Yd, A); torch.cuda.synchronize()
            best = 1e30
            for _ in range(4):  # (best of four runs of 50)
                t0 = time.perf_counter()
                for _ in range(50): hh.fit_device(Xd, Yd, A, out=out)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 50 * 1e6)
            return best
        g = gpu_us(h); ga = gpu_us(ha)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 0.2: one.plsr(X, Y, A); n += 1
        c = (time.perf_counter() - t0) / n * 1e6
        line = "N=%6d K=%5d M=%d A=%d   GPU %8.1f us (AUTO %8.1f)   CPU one core %9.1f us   x%6.1f (x%6.1f)%s" % (N, K, M, A, g, ga, c, c / g, c / ga, "   <-- GPU slower" if min(g, ga) > c else "")
        print(line, flush=True); lines.append(line)
if len(sys.argv) > 1: open(sys.argv[1], "w").write("\n".join(lines) + "\n")
