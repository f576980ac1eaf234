"""Randomised parity sweep beyond the fixed cases of tests/: random shapes, response counts, storage types and plans
against the CPU oracle on the same inputs.  100 cases with random handle options run in the -m gpu suite
(tests/test_gpu_fuzz.py); longer sweeps by hand on an MI355X:
    python tools/fuzz_parity.py [cases] [seed] [options]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pls_amd
from oracle import pls_oracle as po

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
random_options = len(sys.argv) > 3 and sys.argv[3] == "options"  # also draw handle options per case
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
one = po.OracleLib(omp=True)
h = pls_amd.Handle()
worst = {}
bad = 0
widened = 0
t0 = time.time()
for case in range(ncases):
    kind = rng.choice(["tiny", "small", "gram", "wide", "cv"] + (["xwide"] * 5 if os.environ.get("FUZZ_XWIDE") else ["xwide"]))
    M = int(rng.choice([1, 1, 2, 3, 4, 8]))
    if kind == "tiny":
        N = int(rng.integers(1, 1025)); S = 16 // ((N + 63) // 64); K = int(rng.integers(1, max(2, 26 * S + 1)))  # (single-launch fits: 1..8 responses)
    elif kind == "small":
        N = int(rng.integers(2, 3000)); K = int(rng.integers(1, 300))
    elif kind == "gram":
        N = int(rng.integers(4096, 40000)); K = int(rng.integers(1, 700))
    elif kind == "wide":
        N = int(rng.integers(64, 4000)); K = int(rng.integers(1025, 5000))
    elif kind == "xwide":  # row-pack tiles (K <= 8192 / 16384), the split score kernel and the many-workgroup update beyond
        N = int(rng.integers(8, 1500)); K = int(rng.choice([rng.integers(4097, 8193), rng.integers(8193, 16385), rng.integers(16385, 40000)]))
        if 8192 < K <= 16384 and rng.integers(0, 2) == 0: N = int(rng.integers(4096, 5000))  # (the KERNEL plan fuses there from 4096 rows on)
    else:
        N = int(rng.integers(5, 400)); K = int(rng.integers(1, 120))
    A = int(rng.integers(1, min(K, 12) + 1))
    A = max(1, min(A, N - 1))  # beyond the rank of X the reference's results are inf / NaN (src/pls.cpp:427-428): nothing to compare
    f32 = bool(rng.integers(0, 4) == 0)
    X = one.synth_x(case * 7919, N, K); Y = one.synth_y(case * 7919, N, M)
    cond = 0.0
    if f32:
        X = np.asfortranarray(X.astype(np.float32).astype(np.float64)); Y = np.asfortranarray(Y.astype(np.float32).astype(np.float64))
    dt = torch.float32 if f32 else torch.float64
    # layout: half of the cases in the reference's own (ld = rows: odd N leaves columns unaligned), sometimes with extra padding
    # and a base pointer off the 16-byte grid; the rest in 16-byte aligned columns
    if rng.integers(0, 2) == 0:
        ldx, off = N + int(rng.integers(0, 4)), int(rng.integers(0, 4))
        def place(a, cols):
            flat = torch.full((cols * ldx + off + 8,), float("nan"), dtype=dt, device="cuda")
            v = flat[off:off + cols * ldx].view(cols, ldx)[:, :N].t()
            v.copy_(torch.from_numpy(a).to(dt))
            return v
        Xd, Yd = place(X, K), place(Y, M)
    else:
        Xd = pls_amd.as_colmajor(torch.from_numpy(X).to(dt).cuda()); Yd = pls_amd.as_colmajor(torch.from_numpy(Y).to(dt).cuda())
    tol = 5e-5 if f32 else 1e-10  # fp64: the north star's bar on the coefficients
    if random_options:  # every combination must give the same model: layout of the copy, unfused kernels, deferred write-back, grids
        h.set_option(pls_amd.OPT_WORK_LAYOUT, int(rng.integers(0, 2)))
        h.set_option(pls_amd.OPT_FUSE, int(rng.integers(0, 4) != 0))
        h.set_option(pls_amd.OPT_DEFER, int(rng.choice([1, 1, 2, 3, 4])))
        h.set_option(pls_amd.OPT_FUSED_GRID, int(rng.choice([0, 0, 7, 64, 300, 1000])))
    try:
        if kind == "cv":
            ts = int(rng.integers(1, max(2, N // 3))); nf = int(rng.integers(1, 6))
            A = max(1, min(A, N - ts - 1))
            idx = np.stack([rng.permutation(N)[:ts] for _ in range(nf)])
            E = h.cv_folds(Xd, Yd, A, idx).cpu().numpy()
            err = 0.0
            for f in range(nf):
                tr = np.setdiff1d(np.arange(N), idx[f])
                c = one.plsr(np.asfortranarray(X[tr]), np.asfortranarray(Y[tr]), A)
                for nc in (1, A):
                    want = (Y[idx[f]] - X[idx[f]] @ one.coefficients(c["R"], c["Q"], nc)).T
                    err = max(err, float(np.abs(E[:, f * ts:(f + 1) * ts, nc - 1] - want).max() / max(1.0, np.abs(want).max())))
            label = f"cv N={N} K={K} M={M} A={A} ts={ts} f32={f32}"
        else:
            ref = one.plsr(X, Y, A); Bref = one.coefficients(ref["R"], ref["Q"])
            err = 0.0
            for algo in (pls_amd.ALGO_KERNEL, pls_amd.ALGO_NIPALS, pls_amd.ALGO_GRAM, pls_amd.ALGO_AUTO):
                if algo == pls_amd.ALGO_GRAM and K > 2048: continue
                h.set_option(pls_amd.OPT_ALGO, algo)
                o = h.fit_device(Xd, Yd, A)
                err = max(err, po.rel_fro(o["B"].cpu().numpy().astype(np.float64), Bref))
            o2 = h.fit_device(Xd, Yd, A, method=pls_amd.KERNEL_TYPE2) if K <= 4096 else None
            if o2 is not None: err = max(err, po.rel_fro(o2["B"].cpu().numpy().astype(np.float64), Bref))
            label = f"{kind} N={N} K={K} M={M} A={A} f32={f32}"
        # late components of noise-dominated synthetic data are ill-conditioned in B as well: judge against the spread
        # of two CPU routes on the same inputs
        if kind != "cv":
            alt = one.plsr(X, Y, A, nipals=True)
            cond = po.rel_fro(one.coefficients(alt["R"], alt["Q"]), Bref)
            lim = max(tol, 50 * cond)
            if cond > tol / 50:  # the CPU routes themselves disagree beyond the bar: counted, judged against their spread
                widened += 1
        else:
            lim = max(tol, 1e-7)
        if not np.isfinite(err) or err > lim:
            bad += 1
            print("FAIL", label, "err", err, "limit", lim, flush=True)
        key = f"{kind}/{'f32' if f32 else 'f64'}"  # worst figures per storage type
        worst[key] = max(worst.get(key, 0.0), err if np.isfinite(err) else 1e9)
        if kind != "cv" and not f32 and cond <= tol / 50:
            worst["f64, well-conditioned B (limit 1e-10)"] = max(worst.get("f64, well-conditioned B (limit 1e-10)", 0.0), err)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("EXC ", kind, N, K, M, A, f32, repr(e)[:200], flush=True)
    if case % 20 == 19: print(f"{case + 1} cases, {bad} bad, {time.time() - t0:.0f} s, worst {worst}", flush=True)
print("done:", ncases, "cases,", bad, "bad;", widened, "judged against the spread of the CPU routes; worst relative errors", worst)
sys.exit(1 if bad else 0)
