"""Fused-pass launch time against the row count (tile counts around 4096 and 8192), with the column-major
working copy this showed a 10-30 % spread between neighbouring N (DRAM channel effects of 256-byte pieces at
particular column strides); the row-tile-major working copy removed it.  Earlier variants of this script swept the
placement inside a large buffer, the leading dimensions of X and T and the grid size (none of them mattered)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
from pls_amd.model import _ld
from pls_amd import _lib as L
K, M, A = 512, 1, 20
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 1); h.set_option(pls_amd.OPT_ALGO, 1)
big = torch.empty(3 * 1024**3 // 8, dtype=torch.float64, device="cuda:0")
def run(N, off_bytes, grid, label):
    h.set_option(pls_amd.OPT_FUSED_GRID, grid)
    buf = big[off_bytes // 8: off_bytes // 8 + N * K].view(K, N)
    X = buf.t()
    L.check(h._lib.pls_hip_synth_x(h.h, X.data_ptr(), _ld(X), 0, N, K, pls_amd.SEED_DEFAULT, L.F64), h.h)
    Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
    out = h.fit_device(X, Y, A); torch.cuda.synchronize(); h.timing()
    for _ in range(3): h.fit_device(X, Y, A, out=out)
    tm = h.timing()
    us = 1e3 * tm['ms']['fused'] / tm['launches']['fused']
    gb = tm['bytes']['fused'] / tm['launches']['fused'] / 1e9
    print(f"{label}: N={N} off={off_bytes/2**20:.0f} MiB grid={grid or 'auto'}: {us:8.1f} us  {gb/us*1e3:.2f} TB/s", flush=True)
def run_ld(N, ld, label):
    h.set_option(pls_amd.OPT_FUSED_GRID, 0)
    buf = big[: ld * K].view(K, ld)
    X = buf[:, :N].t()
    assert _ld(X) == ld
    L.check(h._lib.pls_hip_synth_x(h.h, X.data_ptr(), _ld(X), 0, N, K, pls_amd.SEED_DEFAULT, L.F64), h.h)
    Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
    out = h.fit_device(X, Y, A); torch.cuda.synchronize(); h.timing()
    for _ in range(3): h.fit_device(X, Y, A, out=out)
    tm = h.timing()
    us = 1e3 * tm['ms']['fused'] / tm['launches']['fused']
    gb = tm['bytes']['fused'] / tm['launches']['fused'] / 1e9
    print(f"{label}: N={N} ld={ld}: {us:8.1f} us  {gb/us*1e3:.2f} TB/s", flush=True)
for nt in [3968, 4032, 4064, 4096, 4128, 4192, 2048, 8128, 8192, 16384]:
    run_ld(nt * 32, nt * 32, f"tiles={nt}")
