"""Wall time per fit at profile level 0 / 1 / 2 (how much the event brackets cost)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
N = int(os.environ.get("TUNE_N", 131072)); K, M, A = 512, 1, 20
h = pls_amd.Handle()
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
out = None
for algo in (1, 0):
    h.set_option(pls_amd.OPT_ALGO, algo)
    for lvl in (0, 1, 2, 0, 1, 2):
        h.set_option(pls_amd.OPT_PROFILE, lvl)
        for _ in range(3): out = h.fit_device(X, Y, A, out=out)
        torch.cuda.synchronize(); h.timing()
        t0 = time.perf_counter()
        for _ in range(20): h.fit_device(X, Y, A, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        h.timing()
        print(f"algo={algo} profile={lvl} N={N}: {dt*1e3:.3f} ms per fit, {A/dt:.0f} comp/s", flush=True)
