"""CPU baselines next to the GPU numbers: the oracle's C restatement of the
reference algorithm (the reference itself needs Eigen and cannot be built) on 1 core (-O3, no -march: mirrors the
reference's Release build) and on all cores (-O3 -march=x86-64-v3 -fopenmp), at config 3 and config 4 (fp64: the
reference has no fp32), config 5 extrapolated linearly in N*K*(1+2A) from those; and the wall times of the two small
reference examples (configs 1, 2) on CPU and GPU."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pls_oracle as po

one, omp = po.OracleLib(omp=False), po.OracleLib(omp=True)
out = {"cores_all": omp.num_threads(), "note": "restatement of the reference algorithm (src/pls.cpp:390-437), Eigen unavailable; "
       "1 core = -O3 without -march/OpenMP like the reference's Release build"}

def timed(lib, X, Y, A, reps=1):
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); lib.plsr(X, Y, A); best = min(best, time.perf_counter() - t0)
    return best

for name, (N, K, M, A) in (("C3", (1 << 20, 512, 1, 20)), ("C4_fp64", (131072, 4096, 8, 50))):
    X = omp.synth_x(0, N, K); Y = omp.synth_y(0, N, M)
    t1 = timed(one, X, Y, A); tn = timed(omp, X, Y, A)
    passes = 1 + 2 * A
    out[name] = {"N": N, "K": K, "M": M, "A": A, "one_core_s": round(t1, 3), "one_core_components_per_s": round(A / t1, 3),
                 "one_core_GBps": round(passes * N * K * 8 / t1 / 1e9, 2), "all_core_s": round(tn, 3),
                 "all_core_components_per_s": round(A / tn, 3), "all_core_GBps": round(passes * N * K * 8 / tn / 1e9, 2)}
    print(name, out[name], flush=True)
    del X, Y
# config 5 (16,777,216 x 1,024, m = 4, A = 20: 137 GB of X): extrapolated from the streaming rates above
N, K, M, A = 16777216, 1024, 4, 20
traffic = (1 + 2 * A) * N * K * 8
out["C5_extrapolated"] = {"N": N, "K": K, "M": M, "A": A, "X_traffic_GB": round(traffic / 1e9, 1),
                          "one_core_s": round(traffic / 1e9 / out["C3"]["one_core_GBps"], 1),
                          "all_core_s": round(traffic / 1e9 / out["C3"]["all_core_GBps"], 1),
                          "note": "EXTRAPOLATED linearly in N*K*(1+2A) from the C3 streaming rates; not run (137 GB of host memory per fit)"}
out["C5_extrapolated"]["one_core_components_per_s"] = round(A / out["C5_extrapolated"]["one_core_s"], 4)
out["C5_extrapolated"]["all_core_components_per_s"] = round(A / out["C5_extrapolated"]["all_core_s"], 4)
# configs 1, 2: the reference's example data
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
import torch, pls_amd
h = pls_amd.Handle()
for name, fx, fy, A in (("C1_toy", "toyX.csv", "toyY.csv", 2), ("C2_nir", "nir.csv", "octane.csv", 10)):
    X = one.z_scores(po.read_csv(os.path.join(DATA, fx))); Y = one.z_scores(po.read_csv(os.path.join(DATA, fy)))
    tc = timed(one, X, Y, A, reps=200)
    Xd = pls_amd.as_colmajor(torch.from_numpy(X).cuda()); Yd = pls_amd.as_colmajor(torch.from_numpy(Y).cuda())
    o = h.fit_device(Xd, Yd, A); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): h.fit_device(Xd, Yd, A, out=o)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(50): h.fit_host(X, Y, A)
    th = (time.perf_counter() - t0) / 50
    out[name] = {"N": X.shape[0], "K": X.shape[1], "M": Y.shape[1], "A": A, "cpu_one_core_us": round(tc * 1e6, 1),
                 "gpu_device_resident_us": round(tg * 1e6, 1), "gpu_host_memory_entry_us": round(th * 1e6, 1),
                 "gpu_over_cpu": round(tg / tc, 1)}
    print(name, out[name], flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)
