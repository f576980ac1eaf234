"""A few KERNEL_TYPE2 fits on config 3 (XX = X^T X on the matrix cores + the K-sized loop), for profiling the
SYRK:  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -- python3 tools/syrk_run.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
N, K, M, A = 1 << 20, 512, 1, 20
dt = torch.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else torch.float64
h = pls_amd.Handle()
X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT, dtype=dt); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT, dtype=dt)
out = None
for _ in range(4):
    out = h.fit_device(X, Y, A, method=pls_amd.KERNEL_TYPE2, out=out)
torch.cuda.synchronize()
