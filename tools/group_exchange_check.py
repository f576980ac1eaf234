"""Functional check of the device-side exchange of pls_hip_group (PLS_HIP_GROUP_EXCHANGE=device) with VIRTUAL members on one
GPU: needs as many hardware queues as members, so run it in a process of its own with GPU_MAX_HW_QUEUES >= members.
For every member count: fits under three plans + KERNEL_TYPE2 against the single-handle fit of the same data, the
members' W, P, Q, R, B bit-identical (checked inside pls_hip_group_fit), predict and SSE through the group.
Prints "exchange check ok" and exits 0, or raises.   python tools/group_exchange_check.py [members ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("PLS_HIP_GROUP_EXCHANGE") == "device", "set PLS_HIP_GROUP_EXCHANGE=device"
import numpy as np
import torch
import pls_amd
from oracle import pls_oracle as po

ora = po.OracleLib()
members = [int(a) for a in sys.argv[1:]] or [2, 3, 4]
h = pls_amd.Handle()
for n in members:
    g = pls_amd.Group([0] * n)
    assert g.exchange == "device", "the self-test of the device-side exchange failed (GPU_MAX_HW_QUEUES >= members?)"
    for (N, K, M, A, algo, method) in ((4096, 64, 1, 6, 0, 0), (4098, 96, 3, 7, 1, 0), (10, 15, 2, 2, 0, 0), (3001, 1300, 2, 5, 1, 0),
                                       (3000, 130, 2, 6, 2, 0), (2000, 200, 1, 5, 0, 1), (2000, 300, 2, 4, 0, 1), (5, 40, 1, 3, 1, 0)):
        Xh, Yh = ora.synth_x(0, N, K), ora.synth_y(0, N, M)
        g.set_option(pls_amd.OPT_ALGO, algo)
        h.set_option(pls_amd.OPT_ALGO, algo)
        X, Y = g.upload(Xh), g.upload(Yh)
        for rep in range(3):  # repeated collectives: both parities of the inbox, sequence numbers carried across fits
            out = g.fit(X, Y, A, method=method)
            if out["T"] is not None and rep < 2:
                g.free(out["T"])
        one = h.fit_device(torch.from_numpy(Xh).cuda(), torch.from_numpy(Yh).cuda(), A, method=method)
        err = po.rel_fro(out["B"], one["B"].cpu().numpy())
        assert err < 1e-11, (n, N, K, M, A, algo, method, err)
        if method == 0:
            T = g.download(out["T"])
            s = po.sign_align(one["W"].cpu().numpy(), out["W"])
            assert po.rel_fro(T * s, one["T"].cpu().numpy()) < 1e-9
            sse = g.model_sse(X, Y, out["R"], out["Q"])
            assert np.allclose(sse[:, A - 1], po.explained_variance(Xh, Yh, out["R"], out["Q"], A)[1], rtol=1e-8)
            g.free(out["T"])
        g.free(X); g.free(Y)
    g.close()
    print(f"members {n}: ok", flush=True)
print("exchange check ok")
