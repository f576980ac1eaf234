"""Sweep PLS_HIP_OPT_FUSED_GRID for the fused pass on BASELINE config 3 and print per-launch times.
Usage (on the GPU box): python tools/tune_fused.py [grid ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pls_amd

N, K, M, A = int(os.environ.get('TUNE_N', 1 << 20)), 512, 1, 6
h = pls_amd.Handle()
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT)
Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
h.set_option(pls_amd.OPT_PROFILE, 1)
grids = [int(g) for g in sys.argv[1:]] or [256, 512, 1024, 2048]
out = None
import itertools
for algo, var, rep in itertools.product((1, 0), (0,), (0, 1)):
    h.set_option(pls_amd.OPT_ALGO, algo)
    for g in grids:
        h.set_option(pls_amd.OPT_FUSED_GRID, g)
        out = h.fit_device(X, Y, A, out=out)
        torch.cuda.synchronize()
        h.timing()
        for _ in range(3):
            h.fit_device(X, Y, A, out=out)
        tm = h.timing()
        n = tm["launches"]["fused"]
        ms = tm["ms"]["fused"] / n
        gb = tm["bytes"]["fused"] / n / ms / 1e6
        if algo:  # time of a read+write launch alone: remove the read-only first pass
            pass
        print(f"var={var} algo={'nipals' if algo else 'kernel'} grid={g:5d} fused avg {ms:.4f} ms  {gb:.0f} GB/s  small/fit {tm['ms']['small']/tm['fits']:.3f} ms  fit {tm['fit_ms']/tm['fits']:.3f} ms", flush=True)
