import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pls_amd
h0 = pls_amd.Handle()
for (N, K, A, dt) in ((200000, 26, 6, torch.float64), (3001, 77, 7, torch.float32), (65537, 50, 9, torch.float32), (3001, 77, 7, torch.float32)):
    X = h0.synth_x(3, N, K, 11, dtype=dt); Y = h0.synth_y(3, N, 1, 11, dtype=dt)
    for mode in ("1", "0"):
        os.environ["PLS_HIP_RESIDENT"] = mode
        h = pls_amd.Handle()
        out = h.fit_device(X, Y, A); h.synchronize()
        ts = []
        t00 = time.perf_counter()
        for _ in range(100):
            t0 = time.perf_counter(); h.fit_device(X, Y, A, out=out); ts.append((time.perf_counter() - t0) * 1e6)
        te = time.perf_counter(); h.synchronize(); tsync = (time.perf_counter() - te) * 1e6
        ts = np.array(ts)
        print(N, K, str(dt)[6:], "resident" if mode == "1" else "general ", "enqueue per fit: median %.1f max %.1f (at rep %d) us; total %.1f us per fit; final sync %.1f us; nan %s" % (
            np.median(ts), ts.max(), int(ts.argmax()), (te - t00 + tsync / 1e6) / 100 * 1e6, tsync, bool(torch.isnan(out["B"]).any())), flush=True)
        h.close()
