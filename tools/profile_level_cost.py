"""Wall time per fit with PLS_HIP_OPT_PROFILE = 0 / 1 / 2 (no events / HIP events around the streaming launches /
around every launch) at 1/8 and all of config 3's rows: what the event brackets cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_ALGO, 1)
for N in (131072, 1048576):
    X = h.synth_x(0, N, 512, 1); Y = h.synth_y(0, N, 1, 1)
    out = h.fit_device(X, Y, 20)
    for lvl in (0, 1, 2, 0, 1):
        h.set_option(pls_amd.OPT_PROFILE, lvl)
        for _ in range(3): h.fit_device(X, Y, 20, out=out)
        torch.cuda.synchronize(); h.timing(); t0 = time.perf_counter()
        for _ in range(20): h.fit_device(X, Y, 20, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20; h.timing()
        print(f"N={N} profile={lvl}: {dt*1e3:.3f} ms per fit, {20/dt:.0f} comp/s", flush=True)
    del X, Y, out
