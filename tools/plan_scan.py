"""Scan of shapes for plan-choice anomalies: per shape the time of one fit under the default plan (ALGO_AUTO), the KERNEL plan
with the one-sweep kernels (OPT_FUSE = 1) and with the one-product kernels (OPT_FUSE = 0), NIPALS both ways.  A default that is
more than 15 % slower than the best KERNEL-sequence alternative is flagged.   usage: plan_scan.py [out.txt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd

h = pls_amd.Handle()
def t_fit(X, Y, A, algo, fuse, reps):
    h.set_option(pls_amd.OPT_ALGO, algo); h.set_option(pls_amd.OPT_FUSE, fuse)
    out = h.fit_device(X, Y, A); torch.cuda.synchronize()
    for _ in range(2): h.fit_device(X, Y, A, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): h.fit_device(X, Y, A, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

lines = []
for dt in (torch.float64, torch.float32):
    for M in (1, 4):
        for K in (16, 64, 256, 1024, 3000, 6000, 12000):
            for N in (200, 1000, 5000, 20000, 100000, 400000):
                if N * K * (8 if dt == torch.float64 else 4) > 3e9 or N * K < 20000: continue
                A = min(10, K, N - 1)
                X = h.synth_x(0, N, K, 5, dtype=dt); Y = h.synth_y(0, N, M, 5, dtype=dt)
                reps = 20 if N * K < 5e7 else 5
                r = {"auto": t_fit(X, Y, A, pls_amd.ALGO_AUTO, 1, reps), "kernel_f1": t_fit(X, Y, A, pls_amd.ALGO_KERNEL, 1, reps),
                     "kernel_f0": t_fit(X, Y, A, pls_amd.ALGO_KERNEL, 0, reps), "nipals_f1": t_fit(X, Y, A, pls_amd.ALGO_NIPALS, 1, reps),
                     "nipals_f0": t_fit(X, Y, A, pls_amd.ALGO_NIPALS, 0, reps)}
                bestk = min(r["kernel_f1"], r["kernel_f0"]); bestn = min(r["nipals_f1"], r["nipals_f0"])
                flag = ""
                if r["auto"] > 1.15 * bestk: flag += "  <-- default %.0f %% behind the best KERNEL form" % ((r["auto"] / bestk - 1) * 100)
                if r["nipals_f1"] > 1.15 * bestn: flag += "  <-- NIPALS one-sweep %.0f %% behind one-product" % ((r["nipals_f1"] / bestn - 1) * 100)
                line = "%s N=%7d K=%6d M=%d A=%2d  auto %8.3f  kernel fused %8.3f one-product %8.3f  nipals fused %8.3f one-product %8.3f ms%s" % (
                    "f64" if dt == torch.float64 else "f32", N, K, M, A, r["auto"], r["kernel_f1"], r["kernel_f0"], r["nipals_f1"], r["nipals_f0"], flag)
                print(line, flush=True); lines.append(line)
                del X, Y
if len(sys.argv) > 1: open(sys.argv[1], "w").write("\n".join(lines) + "\n")
