"""numpy restatement of the tjhladish/PLS fit/predict path + ctypes loader for pls_oracle.c.

TEST INFRASTRUCTURE ONLY (see oracle/pls_oracle.c header): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by pls_amd/.

PARITY UNPINNED: the reference holds no golden vectors for this path and cannot be built
here (Eigen absent); these functions restate /root/reference/src/pls.cpp line by line and are
cross-checked against the C restatement and scikit-learn in tests/test_oracle.py.

The numpy functions are written independently of the C file (different eigen-solver:
LAPACK eigh instead of Jacobi; BLAS products instead of loops) so agreement between the two
is evidence, not tautology.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------------------
# numpy restatement
# --------------------------------------------------------------------------------------


def colwise_z_scores(X: np.ndarray) -> np.ndarray:
    """src/pls.cpp:69-111 + src/main.cpp:24-25: column mean, N-1 stdev, (x-mean)/sd.
    Constant columns divide by the unguarded stdev (:103) -> NaN, as the reference does."""
    X = np.asarray(X, dtype=np.float64)
    mean = X.mean(axis=0)
    sst = ((X - mean) ** 2).sum(axis=0) if X.shape[0] >= 2 else np.zeros(X.shape[1])
    with np.errstate(divide="ignore", invalid="ignore"):
        sd = np.sqrt(sst / (X.shape[0] - 1))
        return (X - mean) / sd


def dominant_direction(XY: np.ndarray) -> np.ndarray:
    """src/pls.cpp:403-411.  M==1: w = XY; else w = XY q with q the dominant eigenvector of
    XY^T XY (:406-408, :113-141).  Sign: largest-|.| entry of q positive."""
    K, M = XY.shape
    if M == 1:
        w = XY[:, 0].copy()
    else:
        lam, V = np.linalg.eigh(XY.T @ XY)
        q = V[:, np.argmax(np.abs(lam))]
        q = q * (1.0 if q[np.argmax(np.abs(q))] >= 0 else -1.0)
        w = XY @ q
    return w / np.sqrt(w @ w)


def plsr(X: np.ndarray, Y: np.ndarray, A: int, method: int = 0):
    """Model::plsr, src/pls.cpp:390-437.  method 0 = KERNEL_TYPE1, 1 = KERNEL_TYPE2.
    Returns dict W,P,Q,R,T (T is None for method 1)."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    N, K = X.shape
    M = Y.shape[1]
    W = np.zeros((K, A)); P = np.zeros((K, A)); R = np.zeros((K, A)); Q = np.zeros((M, A))
    T = np.zeros((N, A)) if method == 0 else None
    XY = X.T @ Y                                   # :396
    XX = X.T @ X if method == 1 else None          # :398
    for i in range(A):                             # :400
        w = dominant_direction(XY)                 # :403-411
        r = w.copy()                               # :412
        for j in range(i):                         # :414-416
            r -= (P[:, j] @ w) * R[:, j]
        if method == 0:
            t = X @ r                              # :419
            tt = t @ t                             # :420
            p = X.T @ t                            # :421
            T[:, i] = t                            # :434
        else:
            tt = r @ XX @ r                        # :423
            p = XX.T @ r                           # :424
        p = p / tt                                 # :427
        q = (r @ XY) / tt                          # :428
        XY = XY - np.outer(p, q) * tt              # :429
        W[:, i] = w; P[:, i] = p; Q[:, i] = q; R[:, i] = r   # :430-433
    return dict(W=W, P=P, Q=Q, R=R, T=T)


def plsr_nipals(X: np.ndarray, Y: np.ndarray, A: int):
    """North-star formulation: explicit X <- X - t p^T after every component."""
    Xd = np.array(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    N, K = Xd.shape
    M = Y.shape[1]
    W = np.zeros((K, A)); P = np.zeros((K, A)); R = np.zeros((K, A)); Q = np.zeros((M, A))
    T = np.zeros((N, A))
    for a in range(A):
        w = dominant_direction(Xd.T @ Y)
        t = Xd @ w
        tt = t @ t
        p = Xd.T @ t / tt
        q = Y.T @ t / tt
        Xd -= np.outer(t, p)
        r = w.copy()
        for j in range(a):
            r -= (P[:, j] @ w) * R[:, j]
        W[:, a] = w; P[:, a] = p; Q[:, a] = q; R[:, a] = r; T[:, a] = t
    return dict(W=W, P=P, Q=Q, R=R, T=T)


def coefficients(R: np.ndarray, Q: np.ndarray, c: int | None = None) -> np.ndarray:
    """Model::coefficients, src/pls.cpp:444-447: B = R[:, :c] Q[:, :c]^T."""
    c = R.shape[1] if c is None else c
    return R[:, :c] @ Q[:, :c].T


def fitted_values(X: np.ndarray, B: np.ndarray) -> np.ndarray:
    """Model::fitted_values, src/pls.cpp:449-451."""
    return np.asarray(X, dtype=np.float64) @ B


def explained_variance(X, Y, R, Q, c):
    """src/pls.cpp:453-467: 1 - SSE/SST per response column; also returns SSE."""
    res = Y - fitted_values(X, coefficients(R, Q, c))
    sse = (res ** 2).sum(axis=0)
    sst = ((Y - Y.mean(axis=0)) ** 2).sum(axis=0)
    return 1.0 - sse / sst, sse


# --------------------------------------------------------------------------------------
# synthetic inputs (spec: DESIGN.md "Synthetic inputs"; bit-identical to the C and HIP twins)
# --------------------------------------------------------------------------------------

_U64 = np.uint64
SYN_F = 8
SEED_DEFAULT = 0x504C5301


def _mix64(z):
    z = (np.asarray(z, dtype=_U64) + _U64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
    return z ^ (z >> _U64(31))


def _u24(stream, idx):
    h = _mix64(_U64(stream) ^ np.asarray(idx, dtype=_U64))
    return ((h >> _U64(40)).astype(np.int64) - 8388608).astype(np.float64) * (1.0 / 8388608.0)


def synth_x(row0: int, nrows: int, K: int, seed: int = SEED_DEFAULT) -> np.ndarray:
    with np.errstate(over="ignore"):
        sE, sZ, sL, sA = (int(_mix64(_U64(seed + d))) for d in (0, 1, 2, 5))
        i = (np.arange(nrows, dtype=_U64) + _U64(row0))[:, None]
        k = np.arange(K, dtype=_U64)[None, :]
        ha = _mix64(_U64(sA) ^ k)   # column noise amplitude 2^-(h%4) * (8 + (h/4)%8)/32  (pls_oracle.c, synth_noise_amp)
        amp = np.ldexp(((ha >> _U64(2)) & _U64(7)).astype(np.float64) + 8.0, -5 - (ha & _U64(3)).astype(np.int64))
        X = amp * _u24(sE, i * _U64(K) + k)
        ltab = np.array([-1.0, -0.5, 0.0, 0.5, 1.0])
        for f in range(SYN_F):
            z = _u24(sZ, i * _U64(SYN_F) + _U64(f))
            L = ltab[(_mix64(_U64(sL) ^ (k * _U64(SYN_F) + _U64(f))) % _U64(5)).astype(np.int64)]
            X = X + z * L
    return np.asfortranarray(X)


def synth_y(row0: int, nrows: int, M: int, seed: int = SEED_DEFAULT) -> np.ndarray:
    with np.errstate(over="ignore"):
        sZ, sC, sN = (int(_mix64(_U64(seed + d))) for d in (1, 3, 4))
        i = (np.arange(nrows, dtype=_U64) + _U64(row0))[:, None]
        j = np.arange(M, dtype=_U64)[None, :]
        S = np.zeros((nrows, M))
        for f in range(SYN_F):
            z = _u24(sZ, i * _U64(SYN_F) + _U64(f))
            C = (_mix64(_U64(sC) ^ (j * _U64(SYN_F) + _U64(f))) % _U64(3)).astype(np.float64) - 1.0
            S = S + z * C
        scale = np.ldexp(1.0, -(np.arange(M) % 16))[None, :]
        Y = scale * S + 0.125 * _u24(sN, i * _U64(M) + j)
    return np.asfortranarray(Y)


# --------------------------------------------------------------------------------------
# ctypes binding of pls_oracle.c
# --------------------------------------------------------------------------------------

_i64 = ctypes.c_int64
_dp = ctypes.POINTER(ctypes.c_double)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, _dp, _i64)


def build(force: bool = False) -> None:
    """Compile liboracle.so / liboracle_omp.so if missing (gcc only; no reference sources)."""
    need = force or not all(os.path.exists(os.path.join(_HERE, n))
                            for n in ("liboracle.so", "liboracle_omp.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def _f(a) -> np.ndarray:
    return np.asfortranarray(a, dtype=np.float64)


class OracleLib:
    """Thin wrapper around liboracle.so (or the OpenMP variant)."""

    def __init__(self, omp: bool = False):
        build()
        self.path = os.path.join(_HERE, "liboracle_omp.so" if omp else "liboracle.so")
        self.lib = L = ctypes.CDLL(self.path)
        L.oracle_plsr.restype = ctypes.c_int
        L.oracle_plsr.argtypes = [_dp, _i64, _dp, _i64, _i64, _i64, _i64, _i64, ctypes.c_int,
                                  _dp, _dp, _dp, _dp, _dp, _i64]
        L.oracle_plsr_nipals.restype = ctypes.c_int
        L.oracle_plsr_nipals.argtypes = [_dp, _i64, _dp, _i64, _i64, _i64, _i64, _i64,
                                         _dp, _dp, _dp, _dp, _dp, _i64]
        L.oracle_plsr_sharded.restype = ctypes.c_int
        L.oracle_plsr_sharded.argtypes = [_dp, _i64, _dp, _i64, _i64, _i64, _i64, _i64,
                                          ALLREDUCE_FN, ctypes.c_void_p,
                                          _dp, _dp, _dp, _dp, _dp, _i64]
        L.oracle_coefficients.restype = None
        L.oracle_coefficients.argtypes = [_dp, _dp, _i64, _i64, _i64, _dp]
        L.oracle_xb.restype = None
        L.oracle_xb.argtypes = [_dp, _i64, _i64, _i64, _dp, _i64, _i64, _dp, _i64]
        L.oracle_xty.restype = None
        L.oracle_xty.argtypes = [_dp, _i64, _dp, _i64, _i64, _i64, _i64, _dp]
        L.oracle_xv.restype = None
        L.oracle_xv.argtypes = [_dp, _i64, _i64, _i64, _dp, _dp]
        L.oracle_xtv.restype = None
        L.oracle_xtv.argtypes = [_dp, _i64, _i64, _i64, _dp, _dp]
        L.oracle_colwise_z_scores.restype = None
        L.oracle_colwise_z_scores.argtypes = [_dp, _i64, _i64, _i64, _dp, _i64, _dp, _dp]
        L.oracle_dominant_eigvec_sts.restype = None
        L.oracle_dominant_eigvec_sts.argtypes = [_dp, _i64, _i64, _dp]
        L.oracle_synth_x.restype = None
        L.oracle_synth_x.argtypes = [_dp, _i64, _i64, _i64, _i64, ctypes.c_uint64]
        L.oracle_synth_y.restype = None
        L.oracle_synth_y.argtypes = [_dp, _i64, _i64, _i64, _i64, ctypes.c_uint64]
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_set_f32_storage.restype = None
        L.oracle_set_f32_storage.argtypes = [ctypes.c_int]
        L.oracle_set_compensated.restype = None
        L.oracle_set_compensated.argtypes = [ctypes.c_int]

    # -- fit ---------------------------------------------------------------------------
    def plsr(self, X, Y, A, method=0, nipals=False, f32_storage=False, compensated=False):
        """f32_storage: emulate fp32 storage of the scores (and of the deflated matrix in the NIPALS form) with
        fp64 arithmetic -- BASELINE config 4's mode, which the reference does not have (pls_oracle.c).
        compensated: every sum in error-free (double-double) form: the same operation sequence without the
        N*eps summation error of index-order sums (pls_oracle.c, "Compensated arithmetic")."""
        X, Y = _f(X), _f(Y)
        self.lib.oracle_set_f32_storage(1 if f32_storage else 0)
        self.lib.oracle_set_compensated(1 if compensated else 0)
        try:
            return self._plsr(X, Y, A, method, nipals)
        finally:
            self.lib.oracle_set_f32_storage(0)
            self.lib.oracle_set_compensated(0)

    def _plsr(self, X, Y, A, method, nipals):
        N, K = X.shape
        M = Y.shape[1]
        W = np.zeros((K, A), order="F"); P = np.zeros((K, A), order="F")
        R = np.zeros((K, A), order="F"); Q = np.zeros((M, A), order="F")
        T = np.zeros((N, A), order="F")
        if nipals:
            rc = self.lib.oracle_plsr_nipals(_ptr(X), N, _ptr(Y), N, N, K, M, A,
                                             _ptr(W), _ptr(P), _ptr(Q), _ptr(R), _ptr(T), N)
        else:
            rc = self.lib.oracle_plsr(_ptr(X), N, _ptr(Y), N, N, K, M, A, method,
                                      _ptr(W), _ptr(P), _ptr(Q), _ptr(R), _ptr(T), N)
        if rc:
            raise ValueError(f"oracle_plsr rc={rc}")
        return dict(W=W, P=P, Q=Q, R=R, T=T if method == 0 else None)

    def plsr_sharded(self, Xloc, Yloc, K, M, A, allreduce):
        """allreduce(np_view) -> None reduces the fp64 view in place across ranks."""
        Xloc, Yloc = _f(Xloc), _f(Yloc)
        Nl = Xloc.shape[0]
        ld = max(Nl, 1)
        W = np.zeros((K, A), order="F"); P = np.zeros((K, A), order="F")
        R = np.zeros((K, A), order="F"); Q = np.zeros((M, A), order="F")
        T = np.zeros((ld, A), order="F")

        def _cb(_user, buf, count):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:  # never let an exception cross the C frame
                import traceback
                traceback.print_exc()
                return 7

        cb = ALLREDUCE_FN(_cb)
        rc = self.lib.oracle_plsr_sharded(_ptr(Xloc), ld, _ptr(Yloc), ld, Nl, K, M, A, cb, None,
                                          _ptr(W), _ptr(P), _ptr(Q), _ptr(R), _ptr(T), ld)
        if rc:
            raise RuntimeError(f"oracle_plsr_sharded rc={rc}")
        return dict(W=W, P=P, Q=Q, R=R, T=T[:Nl])

    # -- pieces ------------------------------------------------------------------------
    def coefficients(self, R, Q, c=None):
        R, Q = _f(R), _f(Q)
        K, A = R.shape
        M = Q.shape[0]
        c = A if c is None else c
        B = np.zeros((K, M), order="F")
        self.lib.oracle_coefficients(_ptr(R), _ptr(Q), K, M, c, _ptr(B))
        return B

    def xb(self, X, Bm):
        X, Bm = _f(X), _f(Bm)
        N, K = X.shape
        C = Bm.shape[1]
        out = np.zeros((N, C), order="F")
        self.lib.oracle_xb(_ptr(X), N, N, K, _ptr(Bm), K, C, _ptr(out), N)
        return out

    def xty(self, X, Y):
        X, Y = _f(X), _f(Y)
        N, K = X.shape
        M = Y.shape[1]
        out = np.zeros((K, M), order="F")
        self.lib.oracle_xty(_ptr(X), N, _ptr(Y), N, N, K, M, _ptr(out))
        return out

    def z_scores(self, X):
        X = _f(X)
        N, K = X.shape
        Z = np.zeros((N, K), order="F")
        self.lib.oracle_colwise_z_scores(_ptr(X), N, N, K, _ptr(Z), N, None, None)
        return Z

    def dominant_eigvec(self, S):
        S = _f(S)
        K, M = S.shape
        q = np.zeros(M)
        self.lib.oracle_dominant_eigvec_sts(_ptr(S), K, M, _ptr(q))
        return q

    def synth_x(self, row0, nrows, K, seed=SEED_DEFAULT, out=None):
        X = np.empty((nrows, K), order="F") if out is None else out
        self.lib.oracle_synth_x(_ptr(X), max(nrows, 1), row0, nrows, K, seed)
        return X

    def synth_y(self, row0, nrows, M, seed=SEED_DEFAULT, out=None):
        Y = np.empty((nrows, M), order="F") if out is None else out
        self.lib.oracle_synth_y(_ptr(Y), max(nrows, 1), row0, nrows, M, seed)
        return Y

    def num_threads(self):
        return self.lib.oracle_num_threads()

    def set_num_threads(self, n):
        self.lib.oracle_set_num_threads(int(n))


def read_csv(path: str) -> np.ndarray:
    """read_matrix_file, src/pls.cpp:37-67: comma separated, no header, one row per line."""
    return np.asfortranarray(np.loadtxt(path, delimiter=",", ndmin=2, dtype=np.float64))


# --------------------------------------------------------------------------------------
# comparison helpers shared by the tests (SURVEY.md section 8(c) "Comparison rule")
# --------------------------------------------------------------------------------------


def rel_fro(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    d = np.linalg.norm(a - b)
    n = np.linalg.norm(b)
    return d / n if n > 0 else d


def sign_align(ref_W, got_W):
    """per-component sign: sign of <w_got, w_ref> (W, P, R, T columns and Q columns flip together)."""
    s = np.sign(np.einsum("ka,ka->a", np.asarray(ref_W), np.asarray(got_W)))
    s[s == 0] = 1.0
    return s


def column_errors(ref, got):
    """per-component relative error of the sign-aligned W,P,R,Q (and T when both have it) columns:
    max over the matrices of |col_got*s - col_ref| / |col_ref|."""
    s = sign_align(ref["W"], got["W"])
    names = ["W", "P", "R", "Q"] + (["T"] if ref.get("T") is not None and got.get("T") is not None else [])
    A = np.asarray(ref["W"]).shape[1]
    err = np.zeros(A)
    for nme in names:
        r = np.asarray(ref[nme], dtype=np.float64); g = np.asarray(got[nme], dtype=np.float64) * s
        d = np.linalg.norm(g - r, axis=0) / np.maximum(np.linalg.norm(r, axis=0), 1e-300)
        err = np.maximum(err, d)
    return err
