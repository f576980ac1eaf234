"""Many components: time per component of one fit as A grows (the r recurrence and the p_j^T w products walk a columns).
usage: many_components_scan.py [out.txt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
h = pls_amd.Handle()
lines = []
for algo, name in ((pls_amd.ALGO_KERNEL, "kernel"), (pls_amd.ALGO_NIPALS, "nipals"), (pls_amd.ALGO_GRAM, "gram")):
    h.set_option(pls_amd.OPT_ALGO, algo)
    for (N, K, M) in ((100000, 512, 1), (100000, 512, 4), (5000, 2000, 1), (400000, 64, 1), (2000, 6000, 2)):
        if name == "gram" and K > 2048: continue
        X = h.synth_x(0, N, K, 5, dtype=torch.float64); Y = h.synth_y(0, N, M, 5, dtype=torch.float64)
        prev = None
        for A in (5, 20, 60, 200, 500):
            if A > min(K, N - 1): continue
            out = h.fit_device(X, Y, A); torch.cuda.synchronize()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps): h.fit_device(X, Y, A, out=out)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
            line = "%-6s N=%7d K=%5d M=%d A=%4d   %10.3f ms per fit   %8.1f us per component" % (name, N, K, M, A, ms, ms * 1e3 / A)
            print(line, flush=True); lines.append(line)
if len(sys.argv) > 1: open(sys.argv[1], "w").write("\n".join(lines) + "\n")
