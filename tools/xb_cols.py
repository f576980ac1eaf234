"""out = X B with 5..32 columns (scores T = X R, src/pls.cpp:439-442): the 4 x 4 x 4 MFMA kernel (PLS_HIP_XB4=1, default; 2 = two
workgroups per CU) beside the older kernels (PLS_HIP_XB4=0): time per call (HIP-event family timing) and error against torch.
usage: xb_cols.py [out.txt] [number of shapes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
lines = []
def handle(mode):
    os.environ["PLS_HIP_XB4"] = str(mode)
    h = pls_amd.Handle(); h.set_option(pls_amd.OPT_PROFILE, 1)
    return h
hs = {m: handle(m) for m in (0, 1)}
shapes = [(torch.float64, 1 << 20, 512), (torch.float64, 1048575, 512), (torch.float64, 1 << 20, 500), (torch.float64, 4194304, 64),
          (torch.float64, 262144, 1024), (torch.float32, 1 << 20, 512), (torch.float32, 131072, 4096), (torch.float64, 131072, 4096),
          (torch.float64, 20000, 2000), (torch.float32, 4194304, 64), (torch.float64, 2000, 20000)]
if len(sys.argv) > 2: shapes = shapes[:int(sys.argv[2])]
for dt, N, K in shapes:
    es = 8 if dt == torch.float64 else 4
    X = hs[0].synth_x(0, N, K, 5, dtype=dt)
    for C in (1, 2, 4, 5, 8, 12, 16, 20, 24, 32):
        B = pls_amd.as_colmajor(torch.randn(K, C, dtype=torch.float64, device="cuda"))
        ref = None
        row = "xb %s N=%8d K=%5d C=%3d " % ("f64" if es == 8 else "f32", N, K, C)
        for m, h in hs.items():
            for _ in range(2): out = h.xb(X, B)
            torch.cuda.synchronize(); h.timing()
            for _ in range(8): out = h.xb(X, B)
            tm = h.timing()
            ms = tm['ms']['xb'] / 8
            by = N * K * es + N * C * es
            if ref is None:
                idx = torch.randint(0, N, (4096,), device="cuda")
                idx[-64:] = torch.arange(N - 64, N, device="cuda")
                ref = X[idx].double() @ B
            err = float((out[idx].double() - ref).abs().max() / ref.abs().max())
            row += " | XB4=%d %7.3f ms %.3f of peak err %.1e" % (m, ms, by / (ms * 1e-3) / 8e12, err)
        print(row, flush=True); lines.append(row)
    del X
if len(sys.argv) > 1: open(sys.argv[1], "w").write("\n".join(lines) + "\n")
