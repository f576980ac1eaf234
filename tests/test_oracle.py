"""CPU tests of the oracle itself (no GPU): the C restatement against the golden fixtures, against
the independent numpy restatement and scikit-learn, the kernel form against the NIPALS-deflation
form, the pre-processing quirks, and the synthetic generator twins."""
import glob
import os

import numpy as np
import pytest

from conftest import DATA, GOLDEN


def _cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def _inputs(oracle, po, name, g):
    if name.startswith("toy"):
        return (oracle.z_scores(po.read_csv(os.path.join(DATA, "toyX.csv"))),
                oracle.z_scores(po.read_csv(os.path.join(DATA, "toyY.csv"))))
    if name.startswith("nir"):
        return (oracle.z_scores(po.read_csv(os.path.join(DATA, "nir.csv"))),
                oracle.z_scores(po.read_csv(os.path.join(DATA, "octane.csv"))))
    N, K, M, seed = (int(g[k]) for k in ("N", "K", "M", "seed"))
    X, Y = oracle.synth_x(0, N, K, seed), oracle.synth_y(0, N, M, seed)
    if "f32" in g.files and int(g["f32"]):  # fp32 storage: the generator's values rounded once (config 4 twin)
        X = np.asfortranarray(X.astype(np.float32).astype(np.float64))
        Y = np.asfortranarray(Y.astype(np.float32).astype(np.float64))
    return X, Y


@pytest.mark.parametrize("name", _cases())
def test_oracle_reproduces_golden(oracle, po, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    X, Y = _inputs(oracle, po, name, g)
    A = int(g["A"])
    c = oracle.plsr(X, Y, A)
    assert po.rel_fro(oracle.coefficients(c["R"], c["Q"]), g["B"]) < 1e-13
    for k in "WPQR":
        if k in g.files:  # the config twins pin B, Q and tt only
            assert po.rel_fro(c[k], g[k]) < 1e-12, k
    assert np.allclose((c["T"] ** 2).sum(0), g["tt"], rtol=1e-12)


def test_survey_sanity_values(oracle, po):
    """SURVEY.md appendix B numbers (obtained independently before this oracle was written)."""
    X = oracle.z_scores(po.read_csv(os.path.join(DATA, "toyX.csv")))
    Y = oracle.z_scores(po.read_csv(os.path.join(DATA, "toyY.csv")))
    c = oracle.plsr(X, Y, 2)
    assert np.allclose((c["T"] ** 2).sum(0), [86.182569925094, 22.944173701730], rtol=1e-11)
    B = oracle.coefficients(c["R"], c["Q"])
    assert np.allclose(B[0], [-0.082614951319, 0.002016221174], atol=1e-11)
    assert np.allclose(B[14], [-0.228401452969, 0.030512497217], atol=1e-11)
    ev, sse = po.explained_variance(X, Y, c["R"], c["Q"], 2)
    assert np.allclose(ev, [0.878239740294, 0.031744982578], atol=1e-10)
    assert np.allclose(sse, [1.095842337351, 8.714295156797], atol=1e-10)
    Xn = oracle.z_scores(po.read_csv(os.path.join(DATA, "nir.csv")))
    Yn = oracle.z_scores(po.read_csv(os.path.join(DATA, "octane.csv")))
    cn = oracle.plsr(Xn, Yn, 10)
    Bn = oracle.coefficients(cn["R"], cn["Q"])
    assert abs(np.linalg.norm(Bn) - 0.451470054526) < 1e-10
    assert np.allclose(Bn[:5, 0], [-0.028959013128, -0.039577589256, -0.022805781861, -0.043051749434,
                                   -0.000840420592], atol=1e-10)


@pytest.mark.parametrize("N,K,M,A", [(60, 17, 1, 8), (200, 33, 3, 10), (513, 40, 8, 12)])
def test_three_routes_agree(oracle, po, N, K, M, A):
    X = oracle.synth_x(0, N, K); Y = oracle.synth_y(0, N, M)
    c = oracle.plsr(X, Y, A)
    B = oracle.coefficients(c["R"], c["Q"])
    n = po.plsr(X, Y, A)
    assert po.rel_fro(po.coefficients(n["R"], n["Q"]), B) < 1e-10          # numpy, LAPACK eigh
    for alt in (oracle.plsr(X, Y, A, nipals=True), po.plsr_nipals(X, Y, A), oracle.plsr(X, Y, A, method=1)):
        assert po.rel_fro(oracle.coefficients(alt["R"], alt["Q"]), B) < 1e-10  # X-deflation form, KERNEL_TYPE2
    from sklearn.cross_decomposition import PLSRegression
    Xc, Yc = X - X.mean(0), Y - Y.mean(0)
    cc = oracle.plsr(Xc, Yc, A)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sk = PLSRegression(n_components=A, scale=False, tol=1e-30, max_iter=100000).fit(Xc, Yc)
    assert po.rel_fro(sk.x_rotations_ @ sk.y_loadings_.T, oracle.coefficients(cc["R"], cc["Q"])) < 1e-9
    # invariants: unit-norm w, P^T R = I, orthogonal scores
    assert np.allclose((c["W"] ** 2).sum(0), 1.0, atol=1e-13)
    assert np.allclose(c["P"].T @ c["R"], np.eye(A), atol=1e-9)
    G = c["T"].T @ c["T"]
    assert np.abs(G - np.diag(np.diag(G))).max() < 1e-9 * np.diag(G).max()


def test_dominant_eigvec(oracle, po):
    rng = np.random.default_rng(3)
    for M in (2, 3, 8, 17):
        S = rng.standard_normal((40, M)) * (2.0 ** -np.arange(M))
        q = oracle.dominant_eigvec(S)
        lam, V = np.linalg.eigh(S.T @ S)
        v = V[:, -1] * np.sign(V[np.argmax(np.abs(V[:, -1])), -1])
        assert np.allclose(q, v, atol=1e-12)
        assert q[np.argmax(np.abs(q))] > 0 and abs(np.linalg.norm(q) - 1) < 1e-14


def test_z_scores_and_constant_column(oracle, po):
    X = po.read_csv(os.path.join(DATA, "toyX.csv"))
    Z = oracle.z_scores(X)
    assert np.allclose(Z.mean(0), 0, atol=1e-12) and np.allclose(Z.std(0, ddof=1), 1, atol=1e-12)
    assert np.allclose(Z, po.colwise_z_scores(X), atol=1e-13)
    Xc = X.copy(); Xc[:, 3] = 1.5
    with np.errstate(all="ignore"):
        Zc = oracle.z_scores(Xc)
    assert np.isnan(Zc[:, 3]).all()       # the reference divides by the unguarded stdev (src/pls.cpp:103)
    assert np.isfinite(np.delete(Zc, 3, axis=1)).all()


def test_synth_twins_bit_identical(oracle, po):
    omp = po.OracleLib(omp=True)
    for (r0, n, K, M) in ((0, 64, 9, 3), (999, 257, 31, 8), (1 << 24, 100, 5, 1)):
        a = oracle.synth_x(r0, n, K); b = po.synth_x(r0, n, K); c = omp.synth_x(r0, n, K)
        assert np.array_equal(a, b) and np.array_equal(a, c)
        assert np.array_equal(oracle.synth_y(r0, n, M), po.synth_y(r0, n, M))
    X = oracle.synth_x(0, 4096, 64)
    assert np.array_equal(X[100:200], oracle.synth_x(100, 100, 64))   # shards of one global matrix
    assert abs(X.mean()) < 0.05 and 0.8 < X.std() < 1.6
    assert np.all(X * 2 ** 31 == np.round(X * 2 ** 31))                # dyadic (noise amplitudes down to 2^-8 steps): exact in fp64


def test_oracle_bad_arguments(oracle):
    X = np.zeros((5, 3), order="F"); Y = np.zeros((5, 1), order="F")
    with pytest.raises(ValueError):
        oracle.plsr(X, Y, 4)   # A > K (the reference asserts, src/pls.cpp:345)


REF_DUMP = os.path.join(os.path.dirname(GOLDEN), os.pardir, "oracle", "_ref", "pls_ref_dump")


def _parse_ref_dump(text):
    """print_state of the reference (src/pls.cpp:564-580): sections "P:", "W:", ... of rows of complex entries "(re,im)";
    the driver appends "fitted:" (real entries)."""
    import re
    out, name, rows = {}, None, []
    for line in text.splitlines() + ["end:"]:
        s = line.strip()
        if re.fullmatch(r"[A-Za-z]+:", s):
            if name is not None:
                out[name] = np.array(rows, dtype=np.float64).reshape(len(rows), -1) if rows else np.zeros((0, 0))
            name, rows = s[:-1], []
            continue
        if not s:
            continue
        cplx = re.findall(r"\(([^,()]+),([^,()]+)\)", s)
        if cplx:
            assert all(float(im) == 0.0 for _, im in cplx), "imaginary parts are zero for real input (SURVEY section 0.4)"
            rows.append([float(re_) for re_, _ in cplx])
        else:
            rows.append([float(v) for v in s.split()])
    return out


@pytest.mark.skipif(not os.path.exists(REF_DUMP),
                    reason="oracle/_ref/pls_ref_dump not built: the reference needs a real <Eigen/Dense> "
                           "(`make -C oracle _ref EIGEN_INC=-I...`); parity stays unpinned without it")
@pytest.mark.parametrize("case", ["toy", "nir", "synth-m1", "synth-m3", "toy-type2"])
def test_oracle_against_reference_build(oracle, po, case, tmp_path):
    """The restatement against the REFERENCE ITSELF (its own src/pls.cpp built with Eigen, driven by oracle/ref_dump.cpp)
    on the reference's example files and two synthetic shapes: W, P, R, Q, T per column modulo sign, B and the fitted
    values to 1e-11.  Runs wherever oracle/_ref was built; skipped otherwise."""
    import subprocess
    method, zs = (2 if case.endswith("type2") else 1), 1
    if case.startswith("toy"):
        fx, fy, A = os.path.join(DATA, "toyX.csv"), os.path.join(DATA, "toyY.csv"), 2
        X, Y = oracle.z_scores(po.read_csv(fx)), oracle.z_scores(po.read_csv(fy))
    elif case == "nir":
        fx, fy, A = os.path.join(DATA, "nir.csv"), os.path.join(DATA, "octane.csv"), 10
        X, Y = oracle.z_scores(po.read_csv(fx)), oracle.z_scores(po.read_csv(fy))
    else:
        N, K, M, A = (300, 24, 1, 8) if case == "synth-m1" else (257, 19, 3, 6)
        X, Y, zs = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M), 0
        fx, fy = str(tmp_path / "X.csv"), str(tmp_path / "Y.csv")
        np.savetxt(fx, X, delimiter=",", fmt="%.17g")
        np.savetxt(fy, Y, delimiter=",", fmt="%.17g")
    r = subprocess.run([REF_DUMP, fx, fy, str(A), str(method), str(zs)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = _parse_ref_dump(r.stdout)
    mine = oracle.plsr(X, Y, A, method=method - 1) if method == 2 else oracle.plsr(X, Y, A)
    B = oracle.coefficients(mine["R"], mine["Q"])
    assert po.rel_fro(B, ref["coefficients"]) < 1e-11
    assert po.rel_fro(X @ B, ref["fitted"]) < 1e-11
    s = po.sign_align(ref["W"], mine["W"])
    for k in "WPRQ":
        assert po.rel_fro(mine[k] * s, ref[k]) < 1e-10, k
    if method == 1:
        assert po.rel_fro(mine["T"] * s, ref["T"]) < 1e-10


def test_ref_dump_parser():
    """the parser of the reference's print_state layout, on a hand-written sample (runs everywhere)"""
    txt = "P:\n(1.5,0) (-2,0)\n(0.25,0) (3,0)\nW:\n(1,0)\nT:\n\ncoefficients:\n(7,0)\nfitted:\n1 2\n3 4\n"
    d = _parse_ref_dump(txt)
    assert np.array_equal(d["P"], [[1.5, -2.0], [0.25, 3.0]]) and np.array_equal(d["W"], [[1.0]])
    assert d["T"].size == 0 and np.array_equal(d["fitted"], [[1.0, 2.0], [3.0, 4.0]]) and d["coefficients"][0, 0] == 7.0
