import os, sys
sys.path.insert(0, "/root/repo")
import torch, pls_amd
def handle(mode):
    os.environ["PLS_HIP_XB4"] = str(mode)
    h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 1)
    return h
hs = {m: handle(m) for m in (0, 1, 3)}
for N, K in ((1 << 20, 512), (262144, 1024), (524288, 1024), (131072, 4096)):
    X = hs[0].synth_x(0, N, K, 5, dtype=torch.float64)
    for C in (8, 12, 20):
        B = pls_amd.as_colmajor(torch.randn(K, C, dtype=torch.float64, device="cuda"))
        row = "N=%d K=%d C=%d" % (N, K, C)
        for m, h in hs.items():
            for _ in range(3): out = h.xb(X, B)
            torch.cuda.synchronize(); h.timing()
            for _ in range(10): out = h.xb(X, B)
            tm = h.timing()
            row += " | XB4=%d %.4f ms" % (m, tm['ms']['xb'] / 10)
        print(row, flush=True)
    del X
