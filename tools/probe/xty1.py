import os, sys
sys.path.insert(0, "/root/repo")
import torch, pls_amd
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 1)
for dt in (torch.float64, torch.float32):
    for N, K in ((1 << 20, 512), (131072, 4096), (65536, 512), (16384, 512), (300000, 256), (20000, 2000), (2000, 20000)):
        X = h.synth_x(0, N, K, 5, dtype=dt)
        for M in (1, 2, 4):
            Y = h.synth_y(0, N, M, 5, dtype=dt)
            for _ in range(3): h.xty(X, Y)
            torch.cuda.synchronize(); h.timing()
            for _ in range(10): r = h.xty(X, Y)
            tm = h.timing()
            ref = X.double().t() @ Y.double()
            print("TILE1=%s %s N=%d K=%d M=%d: %.4f ms  err %.1e" % (os.environ.get("PLS_HIP_XTY_TILE1"), str(dt)[6:], N, K, M, tm['ms']['xty'] / 10, float((r - ref).norm() / ref.norm())), flush=True)
        del X
