"""Per-fit enqueue + completion times of the one-launch fits, loop after loop: looking for stalls (first use of a kernel, scratch).
python tools/resident_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pls_amd
h0 = pls_amd.Handle()
for (N, K, M, A, dt) in ((1025, 26, 3, 5, torch.float64), (2000, 30, 2, 5, torch.float64), (1025, 26, 3, 5, torch.float64), (3001, 77, 1, 7, torch.float32)):
    X = h0.synth_x(3, N, K, 11, dtype=dt); Y = h0.synth_y(3, N, M, 11, dtype=dt)
    h = pls_amd.Handle()
    out = h.fit_device(X, Y, A); h.synchronize()
    for loop in range(4):
        ts = []
        for _ in range(100):
            t0 = time.perf_counter(); h.fit_device(X, Y, A, out=out); h.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
        ts = np.array(ts)
        print(N, K, M, str(dt)[6:], "loop", loop, "sync'd per fit: median %.1f max %.1f (rep %d) mean %.1f us" % (np.median(ts), ts.max(), int(ts.argmax()), ts.mean()), flush=True)
    h.close()
