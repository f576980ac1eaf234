"""Where does component_update_kernel spend its time on config 4 (K=4096, M=8)?  Times the 'small' kernel
family per fit for different power-iteration caps (1 = eigen solve effectively skipped: timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
N, K, M, A = 131072, 4096, 8, 50
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 2); h.set_option(pls_amd.OPT_ALGO, 0)
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT, dtype=torch.float32); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT, dtype=torch.float32)
out = None
for iters in (48, 8, 1, 48):
    h.set_option(pls_amd.OPT_POWER_ITERS, iters)
    out = h.fit_device(X, Y, A, out=out); torch.cuda.synchronize(); h.timing()
    for _ in range(2): h.fit_device(X, Y, A, out=out)
    tm = h.timing()
    print(f"power_iters<={iters}: small family {tm['ms']['small']/tm['fits']:.3f} ms per fit over {tm['launches']['small']//tm['fits']} launches", flush=True)
