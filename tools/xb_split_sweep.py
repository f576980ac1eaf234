#!/usr/bin/env python3
"""t = X v (one column) through pls_hip_xb for shapes around the point where the rows alone stop filling the chip:
   python tools/xb_split_sweep.py        (PLS_HIP_XB_SPLIT=0: rows only)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd

h = pls_amd.Handle()
for dt in (torch.float64, torch.float32):
    for N, K in ((8192, 6144), (32768, 6144), (65536, 6144), (87381, 6144), (131072, 6144), (200000, 6144), (262144, 6144), (131072, 2048),
                 (65536, 20000)):
        X = h.synth_x(0, N, K, 1, dtype=dt)
        v = torch.randn(K, 1, dtype=torch.float64, device="cuda")
        for _ in range(3): h.xb(X, v)
        h.synchronize(); t0 = time.perf_counter()
        for _ in range(10): h.xb(X, v)
        h.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
        gb = N * K * X.element_size() / 1e9
        print(f"{str(dt)[6:]} N={N} K={K}: {ms:.3f} ms {gb / ms:.2f} TB/s  split={os.environ.get('PLS_HIP_XB_SPLIT', '1')}", flush=True)
        del X

# several columns of B on short, wide matrices (scores / fitted values)
for N, K, C in ((512, 50000, 4), (512, 50000, 10), (4096, 32768, 4), (2048, 8192, 20)):
    X = h.synth_x(0, N, K, 1)
    Bm = torch.randn(K, C, dtype=torch.float64, device="cuda")
    for _ in range(3): h.xb(X, Bm)
    h.synchronize(); t0 = time.perf_counter()
    for _ in range(10): h.xb(X, Bm)
    h.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"float64 N={N} K={K} C={C}: {ms:.3f} ms  split={os.environ.get('PLS_HIP_XB_SPLIT', '1')}", flush=True)
    del X
