"""AUTO fits of ONE mid-size shape in a loop (for rocprofv3 --kernel-trace --stats: the one-launch kernel's own duration beside the
wall time per fit).  python tools/probe/rg_one.py N K M A [f32] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pls_amd
N, K, M, A = (int(v) for v in sys.argv[1:5])
dt = torch.float32 if "f32" in sys.argv[5:] else torch.float64
reps = next((int(v) for v in sys.argv[5:] if v.isdigit()), 200)
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_AUTO)
X = h.synth_x(3, N, K, 11, dtype=dt); Y = h.synth_y(3, N, M, 11, dtype=dt)
out = h.fit_device(X, Y, A); h.synchronize()
best = 1e30
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(reps // 5): h.fit_device(X, Y, A, out=out)
    h.synchronize(); best = min(best, (time.perf_counter() - t0) / (reps // 5) * 1e6)
print("%d x %d, %d responses, %d components, %s: %.1f us per fit (wall, best of 5 x %d)" % (N, K, M, A, str(dt)[6:], best, reps // 5), flush=True)
h.close()
