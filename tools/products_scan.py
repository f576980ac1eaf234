"""The products behind scores / fitted_values (out = X B, src/pls.cpp:439-451) and the column statistics, over shapes and
column counts: time and achieved fraction of the 8 TB/s HBM peak (algorithmic bytes: X once + the output).
usage: products_scan.py [out.txt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
h = pls_amd.Handle()
lines = []
def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for dt, es in ((torch.float64, 8), (torch.float32, 4)):
    for (N, K) in ((1 << 20, 512), (131072, 4096), (4194304, 64), (20000, 2000), (1048575, 512), (2000, 20000)):
        X = h.synth_x(0, N, K, 5, dtype=dt)
        for C in (1, 4, 8, 20, 64, 200):
            B = pls_amd.as_colmajor(torch.randn(K, C, dtype=torch.float64, device="cuda"))
            t = timed(lambda: h.xb(X, B), 5 if N * K > 1e8 else 20)
            by = N * K * es + N * C * es
            line = "xb  %s N=%8d K=%6d C=%4d  %9.3f ms  %6.2f TB/s = %.2f of peak (X once)" % ("f64" if es == 8 else "f32", N, K, C, t * 1e3, by / t / 1e12, by / t / 8e12)
            if N * K * es > 5e8 and by / t / 8e12 < 0.45: line += "   <-- low"
            print(line, flush=True); lines.append(line)
        Xc = X.clone()
        t = timed(lambda: h.colwise_z_scores(Xc), 5 if N * K > 1e8 else 20)
        by = 3 * N * K * es
        line = "z   %s N=%8d K=%6d          %9.3f ms  %6.2f TB/s = %.2f of peak (read + read + write)" % ("f64" if es == 8 else "f32", N, K, t * 1e3, by / t / 1e12, by / t / 8e12)
        print(line, flush=True); lines.append(line)
        del X, Xc
if len(sys.argv) > 1: open(sys.argv[1], "w").write("\n".join(lines) + "\n")
