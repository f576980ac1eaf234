"""Per-component cost of the in-process group all-reduce (pls_hip_group) with n VIRTUAL members on one GPU: the config-3
matrix row-sharded over n members, fits with A and 2A components, (t(2A) - t(A)) / A per component, next to the same
shape on a single-member group.  The members share the one GPU, so their passes serialise: the streaming part of a
component costs the same as on one member; what the difference to n = 1 shows is the exchange + the extra launches.
    python tools/group_overhead.py [N] [K] [A]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pls_amd
from pls_amd import _lib as L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
A = int(sys.argv[3]) if len(sys.argv) > 3 else 20
M = 1
lib = L.lib()
res = {}
for n in (1, 2, 4, 8):
    g = pls_amd.Group([0] * n)
    g.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_NIPALS)
    X = g.alloc(N, K); Y = g.alloc(N, M)
    for r in range(n):  # every member generates its own rows on the device
        h = ctypes.c_void_p(); L.check(lib.pls_hip_group_handle(g.g, r, ctypes.byref(h)))
        for m, cols, fn in ((X, K, lib.pls_hip_synth_x), (Y, M, lib.pls_hip_synth_y)):
            ptr = ctypes.c_void_p(); ld = ctypes.c_int64(); r0 = ctypes.c_int64(); nr = ctypes.c_int64()
            L.check(lib.pls_hip_matrix_block(m, r, ctypes.byref(ptr), ctypes.byref(ld), ctypes.byref(r0), ctypes.byref(nr)))
            L.check(fn(h, ptr, ld.value, r0.value, nr.value, cols, pls_amd.SEED_DEFAULT, L.F64), h)
        L.check(lib.pls_hip_synchronize(h), h)
    times = {}
    for a in (A, 2 * A):
        out = g.fit(X, Y, a); g.free(out["T"])
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); out = g.fit(X, Y, a); t1 = time.perf_counter() - t0
            g.free(out["T"]); best = min(best, t1)
        times[a] = best
    per = (times[2 * A] - times[A]) / A
    res[n] = per
    print(f"members {n} ({g.exchange} exchange): fit(A={A}) {times[A]*1e3:.3f} ms, fit(A={2*A}) {times[2*A]*1e3:.3f} ms -> {per*1e6:.1f} us per component"
          + (f", {(per - res[1])*1e6:+.1f} us against one member" if n > 1 else ""), flush=True)
    g.free(X); g.free(Y); g.close()
