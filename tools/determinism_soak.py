"""Race screen: every kernel of the library is deterministic (fixed-order reductions, no atomics on data), so any bit
that differs between repeated fits of the same inputs is a synchronisation bug (e.g. an LDS-DMA slab read before it
landed in the SYRK).  Repeats NIPALS / KERNEL / GRAM / KERNEL_TYPE2 fits and compares every output bit for bit.
usage: determinism_soak.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
h = pls_amd.Handle()
cases = [("nipals", pls_amd.ALGO_NIPALS, pls_amd.KERNEL_TYPE1, 262144, 512, 1, 6, torch.float64),
         ("kernel", pls_amd.ALGO_KERNEL, pls_amd.KERNEL_TYPE1, 262144, 512, 2, 6, torch.float64),
         ("gram", pls_amd.ALGO_GRAM, pls_amd.KERNEL_TYPE1, 131072, 512, 1, 8, torch.float64),
         ("type2-f32", pls_amd.ALGO_KERNEL, pls_amd.KERNEL_TYPE2, 131072, 384, 3, 8, torch.float32),
         ("nipals-wide-f32", pls_amd.ALGO_NIPALS, pls_amd.KERNEL_TYPE1, 16384, 4096, 8, 6, torch.float32),
         ("kernel-wide", pls_amd.ALGO_KERNEL, pls_amd.KERNEL_TYPE1, 16384, 2048, 2, 6, torch.float64),
         ("nipals-mid", pls_amd.ALGO_NIPALS, pls_amd.KERNEL_TYPE1, 65536, 1024, 4, 6, torch.float64)]
bad = 0
for name, algo, method, N, K, M, A, dt in cases:
    h.set_option(pls_amd.OPT_ALGO, algo)
    X = h.synth_x(0, N, K, 7, dtype=dt); Y = h.synth_y(0, N, M, 7, dtype=dt)
    ref = {k: v.clone() for k, v in h.fit_device(X, Y, A, method=method).items() if v is not None}
    torch.cuda.synchronize()
    diffs = 0
    for r in range(reps):
        out = h.fit_device(X, Y, A, method=method); torch.cuda.synchronize()
        for k, v in ref.items():
            if k == "T" and method == pls_amd.KERNEL_TYPE2: continue
            if not torch.equal(out[k], v): diffs += 1
    print(f"{name}: {reps} repeats, {diffs} outputs differing", flush=True)
    bad += diffs
    del X, Y
print("RESULT", "ok" if bad == 0 else f"{bad} differences")
sys.exit(0 if bad == 0 else 1)
