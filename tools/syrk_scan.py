"""X^T X on the matrix cores over shapes (the SYRK of the GRAM / KERNEL_TYPE2 plans): launch time by HIP events and the rate on
the tiles it executes (triangular diagonal blocks) against 78.6 TF fp64.   usage: syrk_scan.py [out.txt]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 1); h.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_GRAM)
lines = []
for dt, es in ((torch.float64, 8), (torch.float32, 4)):
    for K in (64, 128, 256, 384, 512, 768, 896, 1024, 1280, 1536, 2048, 4096):
        for gb in (0.5, 4.0):
            N = int(gb * 1e9 / (K * es)) // 64 * 64
            if N < 1024: continue
            X = h.synth_x(0, N, K, 5, dtype=dt); Y = h.synth_y(0, N, 1, 5, dtype=dt)
            out = h.fit_device(X, Y, 3); torch.cuda.synchronize(); h.timing()
            for _ in range(3): h.fit_device(X, Y, 3, out=out)
            tm = h.timing()
            ms = tm["ms"]["xty"] / max(tm["fits"], 1)  # per fit: the SYRK launch (X^T Y on board; a separate X^T Y pass would add to it)
            per_fit = tm["launches"]["xty"] / max(tm["fits"], 1)
            nbk = (K + 127) // 128
            tiles = (nbk * (nbk - 1) // 2) * 64 + nbk * 36  # as executed: full 128-column blocks (a ragged last block computes padding)
            gf = 2.0 * N * tiles * 256 / 1e9
            line = "%s N=%9d K=%5d  SYRK %8.3f ms  %7.1f GF executed  %5.1f TFLOP/s = %.3f of 78.6" % ("f64" if es == 8 else "f32", N, K, ms, gf, gf / ms, gf / ms / 78.6) + ("" if per_fit == 1 else "   (%g launches of the family per fit)" % per_fit)
            print(line, flush=True); lines.append(line)
            del X, Y, out
if len(sys.argv) > 1: open(sys.argv[1], "w").write("\n".join(lines) + "\n")
