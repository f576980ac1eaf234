#!/usr/bin/env python3
"""Time of pls_hip_colwise_z_scores on a resident matrix, out of place:
   python tools/zscore_time.py [N K [f32]]
Traffic: statistics read X once (one sweep) or twice (two passes); the scale pass reads X and writes Z."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import pls_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dt = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else torch.float64
h = pls_amd.Handle()
X = h.synth_x(0, N, K, 1, dtype=dt)
Z = pls_amd.colmajor_empty(N, K, dt, "cuda")
mean = torch.empty(K, dtype=torch.float64, device="cuda"); sd = torch.empty_like(mean)
from pls_amd import _lib as L
from pls_amd.model import _ld


def run(z):
    L.check(h._lib.pls_hip_colwise_z_scores(h.h, X.data_ptr(), _ld(X), N, N, K, h._dt(X), z.data_ptr() if z is not None else 0,
                                            _ld(z) if z is not None else 0, mean.data_ptr(), sd.data_ptr()), h.h)


for what, z in (("statistics only", None), ("statistics + scale", Z), ("... in place", X)):
    for _ in range(3):
        run(z)
    h.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        run(z)
    h.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    gb = N * K * X.element_size() / 1e9
    print(f"{what:20s} N={N} K={K} {str(dt)[6:]}: {ms:.3f} ms  ({gb:.2f} GB per sweep of X; "
          f"one-sweep statistics)")
