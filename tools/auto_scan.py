"""ALGO_AUTO against the explicit KERNEL and GRAM plans: does the cost model pick the faster one?  fp64, one response.
usage: auto_scan.py [out.txt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
h = pls_amd.Handle()
lines = []
def t_fit(X, Y, A, algo, reps):
    h.set_option(pls_amd.OPT_ALGO, algo)
    out = h.fit_device(X, Y, A); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): h.fit_device(X, Y, A, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for dt in (torch.float64, torch.float32):
    for N in (5000, 50000, 500000, 4000000):
        for K in (32, 128, 512, 1024, 2048):
            if N * K > 2.2e9: continue
            for A in (3, 10, 40):
                if A > K: continue
                X = h.synth_x(0, N, K, 5, dtype=dt); Y = h.synth_y(0, N, 1, 5, dtype=dt)
                reps = 3 if N * K > 1e8 else 20
                ta, tk, tg = t_fit(X, Y, A, pls_amd.ALGO_AUTO, reps), t_fit(X, Y, A, pls_amd.ALGO_KERNEL, reps), t_fit(X, Y, A, pls_amd.ALGO_GRAM, reps)
                best = min(tk, tg)
                flag = "   <-- AUTO %.0f %% behind the better plan" % ((ta / best - 1) * 100) if ta > 1.15 * best and ta - best > 0.03 else ""
                line = "%s N=%8d K=%5d A=%2d   auto %8.3f  kernel %8.3f  gram %8.3f ms%s" % ("f64" if dt == torch.float64 else "f32", N, K, A, ta, tk, tg, flag)
                print(line, flush=True); lines.append(line)
                del X, Y
if len(sys.argv) > 1: open(sys.argv[1], "w").write("\n".join(lines) + "\n")
