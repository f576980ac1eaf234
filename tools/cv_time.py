"""Cross-validation of the reference's example data (nir: 60 x 401 -> octane, 10 components), the two calls of its main
(src/main.cpp: leave-one-out = 60 folds; leave-some-out = 10 N = 600 random splits of 18 test rows): all folds in one
launch on the device against one CPU refit per fold (the reference's procedure; oracle restatement on one core)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pls_amd
from oracle import pls_oracle as po

one = po.OracleLib(omp=False)
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
X = one.z_scores(po.read_csv(os.path.join(DATA, "nir.csv"))); Y = one.z_scores(po.read_csv(os.path.join(DATA, "octane.csv")))
N, A = X.shape[0], 10
h = pls_amd.Handle()
Xd = pls_amd.as_colmajor(torch.from_numpy(X).cuda()); Yd = pls_amd.as_colmajor(torch.from_numpy(Y).cuda())
rng = np.random.default_rng(1)
out = {}
for name, idx in (("LOO_60_folds", np.arange(N)[:, None]), ("LSO_600_folds_of_18", np.stack([rng.permutation(N)[:18] for _ in range(10 * N)]))):
    E = h.cv_folds(Xd, Yd, A, idx); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): E = h.cv_folds(Xd, Yd, A, idx)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 20
    nf = idx.shape[0]
    sample = min(nf, 60)
    Eh = E.cpu().numpy()
    trains = [np.setdiff1d(np.arange(N), idx[f]) for f in range(sample)]
    Xs = [np.asfortranarray(X[t]) for t in trains]; Ys = [np.asfortranarray(Y[t]) for t in trains]
    t0 = time.perf_counter()
    fits = [one.plsr(Xs[f], Ys[f], A) for f in range(sample)]
    tc = (time.perf_counter() - t0) / sample * nf
    worst = 0.0
    for f in range(3):
        for nc in (1, A):
            want = (Y[idx[f]] - X[idx[f]] @ one.coefficients(fits[f]["R"], fits[f]["Q"], nc)).T
            worst = max(worst, float(np.abs(Eh[:, f * idx.shape[1]:(f + 1) * idx.shape[1], nc - 1] - want).max()))
    out[name] = {"folds": nf, "gpu_ms": round(tg * 1e3, 3), "cpu_one_core_ms_refits_only": round(tc * 1e3, 2), "speedup": round(tc / tg, 1),
                 "max_residual_difference_checked": worst}
    print(name, out[name], flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)
