"""Fixed cost per launch of the fused pass: average launch duration (HIP events, profile level 1) of the
NIPALS read+write pass at K = 512 fp64 for row counts of 1..64 tiles per workgroup (512 workgroups x 32 rows),
and for leading dimensions padded away from a power of two."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
from pls_amd.model import colmajor_empty, _ld
from pls_amd import _lib as L
K, M, A = 512, 1, 20
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 1); h.set_option(pls_amd.OPT_ALGO, 1)
cases = [(t, 0) for t in (1, 2, 4, 8, 12, 16, 64)] + [(8.0625, 0), (7.9375, 0)]
for tiles, pad in cases:
    N = int(32 * 512 * tiles)
    X = colmajor_empty(N, K, torch.float64, "cuda:0", ld=N + pad)
    L.check(h._lib.pls_hip_synth_x(h.h, X.data_ptr(), _ld(X), 0, N, K, pls_amd.SEED_DEFAULT, L.F64), h.h)
    Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
    out = h.fit_device(X, Y, A); torch.cuda.synchronize(); h.timing()
    for _ in range(3): h.fit_device(X, Y, A, out=out)
    tm = h.timing()
    us = 1e3 * tm['ms']['fused'] / tm['launches']['fused']
    gb = tm['bytes']['fused'] / tm['launches']['fused'] / 1e9
    print(f"tiles/WG={tiles:7.3f} N={N:8d} ld=N+{pad:<5d}: fused launch {us:8.1f} us  {gb/us*1e3:.2f} TB/s  per tile {us/tiles:.1f} us", flush=True)
    del X, Y, out
