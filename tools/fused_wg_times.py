"""Per-workgroup start/end times of the last fused pass of a fit.

Needs a TEMPORARY instrumented build that is not in the tree: fused_pass_kernel takes one more pointer argument and
thread 0 of every workgroup writes {wall_clock64() at entry, at exit, chunks, tiles} to it; the launcher reads the
device address from the environment variable PLS_HIP_TSBUF.  The numbers this produced are recorded in
profiles/r1/fused_wg_times.txt (static tile map vs device work queue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
ts = torch.zeros(4 * 1024, dtype=torch.int64, device="cuda:0")
os.environ["PLS_HIP_TSBUF"] = hex(ts.data_ptr())
import pls_amd
K, M, A = 512, 1, 6
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_ALGO, 1)
for nt in (4096, 4128, 8192, 32768):
    N = nt * 32
    X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
    out = h.fit_device(X, Y, A); out = h.fit_device(X, Y, A, out=out); torch.cuda.synchronize()
    t = ts.cpu().numpy().reshape(-1, 4)[:512]
    t0 = t[:, 0].min()
    st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0   # us (100 MHz)
    nch, ntl = t[:, 2], t[:, 3]
    print(f"tiles={nt}: kernel span {en.max():.1f} us; start spread {st.max():.1f}; end min/median/max {en.min():.1f}/{np.median(en):.1f}/{en.max():.1f}")
    print(f"   chunks per WG min/mean/max {nch.min()}/{nch.mean():.1f}/{nch.max()}; tiles per WG min/mean/max {ntl.min()}/{ntl.mean():.1f}/{ntl.max()}; total tiles {ntl.sum()}")
    print(f"   WG 0-255: tiles mean {ntl[:256].mean():.1f} end mean {en[:256].mean():.1f}; WG 256-511: tiles mean {ntl[256:].mean():.1f} end mean {en[256:].mean():.1f}")
    hw = t[:, 3]
    del X, Y, out
