"""Cost of the dominant-eigenvector solve inside the component update at config 4 (K = 4096, M = 8): average duration
of the 'small' kernel family per update for different caps on the number of squarings (OPT_POWER_ITERS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
N, K, M, A = 131072, 4096, 8, 50
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 2); h.set_option(pls_amd.OPT_ALGO, 0)
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT, dtype=torch.float32); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT, dtype=torch.float32)
out = None
for iters in (48, 1, 2, 4, 6, 8, 12, 16, 24, 48):
    h.set_option(pls_amd.OPT_POWER_ITERS, iters)
    out = h.fit_device(X, Y, A, out=out); torch.cuda.synchronize(); h.timing()
    for _ in range(3): h.fit_device(X, Y, A, out=out)
    tm = h.timing()
    print(f"power_iters<={iters:2d}: small family {1e3*tm['ms']['small']/tm['launches']['small']:.2f} us per launch ({tm['launches']['small']//tm['fits']} launches per fit), fit {tm['fit_ms']/tm['fits']:.3f} ms", flush=True)
