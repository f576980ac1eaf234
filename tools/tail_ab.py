#!/usr/bin/env python3
"""Fits a few shapes and prints a digest of every output, for A/B runs of library switches that must not change a bit
(PLS_HIP_TAIL=0 | 1: the partial rows summed by reduce_partials_kernel or in the tail of the pass -- same order of
additions by construction).  usage: python tools/tail_ab.py [ranks-per-shape ...]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import pls_amd  # noqa: E402

SHAPES = [  # N, K, M, A, algo, dtype
    (131072, 512, 1, 6, pls_amd.ALGO_NIPALS, torch.float64), (131072, 512, 1, 6, pls_amd.ALGO_KERNEL, torch.float64),
    (65537, 300, 3, 5, pls_amd.ALGO_NIPALS, torch.float64), (65537, 300, 3, 5, pls_amd.ALGO_KERNEL, torch.float64),
    (40000, 1000, 4, 5, pls_amd.ALGO_NIPALS, torch.float64), (20001, 4096, 8, 4, pls_amd.ALGO_NIPALS, torch.float32),
    (20001, 4096, 8, 4, pls_amd.ALGO_KERNEL, torch.float32), (300000, 40, 1, 5, pls_amd.ALGO_NIPALS, torch.float64),
    (9000, 7000, 1, 4, pls_amd.ALGO_KERNEL, torch.float64), (2048, 64, 1, 4, pls_amd.ALGO_NIPALS, torch.float64),
]


def main():
    torch.cuda.set_device(0)
    h = pls_amd.Handle()
    for N, K, M, A, algo, dt in SHAPES:
        h.set_option(pls_amd.OPT_ALGO, algo)
        X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT, dtype=dt)
        Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT, dtype=dt)
        out = h.fit_device(X, Y, A)
        h.synchronize()
        dig = hashlib.sha256()
        for k in "WPQRTB":
            dig.update(out[k].cpu().numpy().tobytes())
        print(N, K, M, A, algo, str(dt)[6:], dig.hexdigest()[:24], flush=True)
        del X, Y, out


if __name__ == "__main__":
    main()
