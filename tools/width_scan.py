"""Column counts either side of the tile-shape boundaries of the one-sweep passes (512 / 1024 / 2048 / 4096 / 8192 / 16384): time per
component of a 20-component fit on ~3 GB matrices and the effective rate (KERNEL: X read once per component, NIPALS: read + written)
as a fraction of the 8 TB/s HBM peak.   usage: width_scan.py [out.txt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
h = pls_amd.Handle()
lines = []
for dt, es in ((torch.float64, 8), (torch.float32, 4)):
    for K in (66, 80, 96, 100, 130, 160, 190, 200):
        N = int(3e9 / (K * es)) // 64 * 64
        A = 20
        X = h.synth_x(0, N, K, 5, dtype=dt); Y = h.synth_y(0, N, 1, 5, dtype=dt)
        row = "%s K=%6d N=%8d " % ("f64" if es == 8 else "f32", K, N)
        for algo, name, passes in ((pls_amd.ALGO_KERNEL, "kernel", 1), (pls_amd.ALGO_NIPALS, "nipals", 2)):
            h.set_option(pls_amd.OPT_ALGO, algo)
            out = h.fit_device(X, Y, A); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2): h.fit_device(X, Y, A, out=out)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 2 * 1e3
            rate = passes * N * K * es * A / (ms * 1e-3) / 8e12
            row += "  %s %8.2f ms/fit %6.0f comp/s  %.2f of peak" % (name, ms, A / ms * 1e3, rate)
            del out
        print(row, flush=True); lines.append(row)
        del X, Y
if len(sys.argv) > 1: open(sys.argv[1], "w").write("\n".join(lines) + "\n")
