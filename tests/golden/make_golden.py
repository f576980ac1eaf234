"""Regenerates tests/golden/*.npz.  Run from the repo root:  python tests/golden/make_golden.py

The reference holds no golden vectors and cannot be built here (Eigen absent), so these
fixtures are produced by the C restatement in oracle/pls_oracle.c and are only written when two
independent routes agree with it: the numpy restatement (oracle/pls_oracle.py) and, for m == 1 or
well-separated directions, scikit-learn's NIPALS PLSRegression(scale=False, tol=1e-30).
Inputs: the reference's own example data (tests/golden/data/*.csv, copied verbatim from
/root/reference/{toyX,toyY,nir,octane}.csv) z-scored as src/main.cpp:24-25 does, and seeded
synthetic matrices (generator spec in DESIGN.md) identified by (N, K, M, seed) only.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pls_oracle as po  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")

# (name, N, K, M, A) -- ragged shapes on purpose (non-multiples of every tile size)
SYNTH_CASES = [
    ("s_9x7_m1", 9, 7, 1, 5),
    ("s_9x7_m2", 9, 7, 2, 4),
    ("s_1000x7_m4", 1000, 7, 4, 7),
    ("s_1000x64_m1", 1000, 64, 1, 20),
    ("s_1000x64_m4", 1000, 64, 4, 12),
    ("s_1000x513_m8", 1000, 513, 8, 8),
    ("s_4097x64_m2", 4097, 64, 2, 16),
    ("s_4097x513_m1", 4097, 513, 1, 8),
    ("s_4097x7_m8", 4097, 7, 8, 7),
    ("s_2048x512_m1", 2048, 512, 1, 20),   # small-N twin of BASELINE config 3
    ("s_2048x1024_m4", 2048, 1024, 4, 6),  # small-N twin of config 5
]


def sklearn_B(X, Y, A):
    from sklearn.cross_decomposition import PLSRegression
    sk = PLSRegression(n_components=A, scale=False, tol=1e-30, max_iter=200000).fit(X, Y)
    return sk.x_rotations_ @ sk.y_loadings_.T


def fit_and_check(ora, X, Y, A, name, tol=1e-9):
    c = ora.plsr(X, Y, A)
    B = ora.coefficients(c["R"], c["Q"])
    n = po.plsr(X, Y, A)
    e_np = po.rel_fro(po.coefficients(n["R"], n["Q"]), B)
    c2 = ora.plsr(X, Y, A, nipals=True)
    e_ni = po.rel_fro(ora.coefficients(c2["R"], c2["Q"]), B)
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            # scikit-learn always centres X and Y; compare on centred copies of the same data
            Xc, Yc = X - X.mean(0), Y - Y.mean(0)
            cc = ora.plsr(Xc, Yc, A)
            e_sk = po.rel_fro(sklearn_B(Xc, Yc, A), ora.coefficients(cc["R"], cc["Q"]))
    print(f"{name:18s} B rel-err vs numpy {e_np:.1e}  vs nipals-form {e_ni:.1e}  vs sklearn {e_sk:.1e}")
    assert e_np < tol and e_ni < tol and e_sk < 1e-7, name
    # Per-component conditioning: once the latent structure is exhausted the directions of the
    # noise components are nearly degenerate and W/P/R/Q/T columns (not B) of two correct fp64
    # implementations drift apart by a growing factor per component.  Record how far the two
    # independent CPU routes are from the C oracle, column by column; the GPU tests require the
    # HIP path to agree with the oracle to max(1e-10, 20 x this).
    c["col_err"] = np.maximum(po.column_errors(c, n), po.column_errors(c, c2))
    return c, B


def main():
    ora = po.OracleLib()
    # --- reference example data ---------------------------------------------------------
    for name, fx, fy, A in (("toy_A2", "toyX.csv", "toyY.csv", 2), ("nir_A10", "nir.csv", "octane.csv", 10)):
        X = ora.z_scores(po.read_csv(os.path.join(DATA, fx)))
        Y = ora.z_scores(po.read_csv(os.path.join(DATA, fy)))
        c, B = fit_and_check(ora, X, Y, A, name)
        ev = np.array([po.explained_variance(X, Y, c["R"], c["Q"], k)[0] for k in range(1, A + 1)])
        sse = np.array([po.explained_variance(X, Y, c["R"], c["Q"], k)[1] for k in range(1, A + 1)])
        np.savez(os.path.join(HERE, name + ".npz"), A=A, W=c["W"], P=c["P"], Q=c["Q"], R=c["R"], T=c["T"],
                 B=B, explained_variance=ev, SSE=sse, tt=(c["T"] ** 2).sum(0), col_err=c["col_err"])
    # --- seeded synthetic shapes ----------------------------------------------------------
    for name, N, K, M, A in SYNTH_CASES:
        X = ora.synth_x(0, N, K)
        Y = ora.synth_y(0, N, M)
        c, B = fit_and_check(ora, X, Y, A, name)
        out = dict(N=N, K=K, M=M, A=A, seed=po.SEED_DEFAULT, W=c["W"], P=c["P"], Q=c["Q"], R=c["R"], B=B,
                   tt=(c["T"] ** 2).sum(0), col_err=c["col_err"], x_checksum=float(np.abs(X).sum()), y_checksum=float(np.abs(Y).sum()))
        if N <= 1000:
            out["T"] = c["T"]
        np.savez(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    main()
