"""X^T Y at config 4's shape (131,072 x 4,096 fp32, M = 8): HIP-event time of the xty family per call.
(4 columns per workgroup of the 8-response tile: the 8- and 16-column forms measured in round 2 are deleted)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
N, K, M = 131072, 4096, 8
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 1)
for dt in (torch.float32, torch.float64):
    X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT, dtype=dt); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT, dtype=dt)
    for _ in range(3): h.xty(X, Y)
    torch.cuda.synchronize(); h.timing()
    for _ in range(10): h.xty(X, Y)
    tm = h.timing()
    ms = tm['ms']['xty'] / tm['launches']['xty']
    gb = tm['bytes']['xty'] / tm['launches']['xty'] / 1e9
    ref = (X.double().t() @ Y.double())
    err = float((h.xty(X, Y) - ref).norm() / ref.norm())
    print(f"{str(dt):14s} KC8=4: {ms:.4f} ms per launch, {gb/ms*1e3:.0f} GB/s = {gb/ms/8:.3f} of peak, rel err {err:.1e}", flush=True)
    del X, Y
