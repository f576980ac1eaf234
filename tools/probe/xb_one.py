import os, sys
sys.path.insert(0, "/root/repo")
import torch, pls_amd
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 1)
N, K = 1 << 20, 512
X = h.synth_x(0, N, K, 5, dtype=torch.float64)
for C in (8, 20):
    B = pls_amd.as_colmajor(torch.randn(K, C, dtype=torch.float64, device="cuda"))
    for _ in range(3): out = h.xb(X, B)
    torch.cuda.synchronize(); h.timing()
    for _ in range(10): out = h.xb(X, B)
    tm = h.timing()
    ref = X[:4096].double() @ B
    err = float((out[:4096] - ref).abs().max() / ref.abs().max())
    print("mode %s XB4=%s C=%d: %.4f ms  err %.2e" % (os.environ.get("PLS_HIP_XB4_MODE"), os.environ.get("PLS_HIP_XB4"), C, tm['ms']['xb'] / 10, err), flush=True)
