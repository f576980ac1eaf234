"""Configs 1 and 2 (the reference's example data): one fit as ONE launch (tiny_kernels.hpp) against the three-launches-per-
component plan (PLS_HIP_TINY=0) and the CPU restatement on one core.  Per fit: back-to-back (asynchronous launches,
200 fits, one synchronisation) and as a latency (synchronised after every fit)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pls_amd
from oracle import pls_oracle as po

one = po.OracleLib(omp=False)
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
h = pls_amd.Handle()
out = {}
for name, fx, fy, A in (("C1_toy", "toyX.csv", "toyY.csv", 2), ("C2_nir", "nir.csv", "octane.csv", 10)):
    X = one.z_scores(po.read_csv(os.path.join(DATA, fx))); Y = one.z_scores(po.read_csv(os.path.join(DATA, fy)))
    best = 1e30
    for _ in range(300):
        t0 = time.perf_counter(); ref = one.plsr(X, Y, A); best = min(best, time.perf_counter() - t0)
    row = {"N": X.shape[0], "K": X.shape[1], "M": Y.shape[1], "A": A, "cpu_one_core_us": round(best * 1e6, 1)}
    Xd = pls_amd.as_colmajor(torch.from_numpy(X).cuda()); Yd = pls_amd.as_colmajor(torch.from_numpy(Y).cuda())
    for tag, env in (("single_launch", "1"), ("three_launches_per_component", "0")):
        os.environ["PLS_HIP_TINY"] = env  # (read when a handle is created)
        h = pls_amd.Handle()
        o = h.fit_device(Xd, Yd, A); torch.cuda.synchronize()
        Bref = one.coefficients(ref["R"], ref["Q"])
        err = float(np.abs(o["B"].cpu().numpy() - Bref).max())
        t0 = time.perf_counter()
        for _ in range(200): h.fit_device(Xd, Yd, A, out=o)
        torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 200
        t0 = time.perf_counter()
        for _ in range(200):
            h.fit_device(Xd, Yd, A, out=o); torch.cuda.synchronize()
        tl = (time.perf_counter() - t0) / 200
        h.fit_host(X, Y, A)  # buffers of the host entry exist
        t0 = time.perf_counter()
        for _ in range(50): h.fit_host(X, Y, A)
        th = (time.perf_counter() - t0) / 50
        row[tag] = {"back_to_back_us": round(tb * 1e6, 1), "latency_us": round(tl * 1e6, 1), "host_memory_entry_us": round(th * 1e6, 1),
                    "B_err_vs_cpu": err}
    del os.environ["PLS_HIP_TINY"]
    out[name] = row
    print(name, row, flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)
