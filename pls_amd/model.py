"""Python host mirror of the reference's `PLS::Model` fit / predict surface.

Same names, argument meaning and defaults as /root/reference/include/PLS/pls.h:184-266, over
the C-ABI of include/pls_hip.h.  Differences, all at the type level only:
  * matrices are torch CUDA tensors (device path, zero copy) or numpy arrays (host path,
    copied in and out by the library) instead of Eigen matrices;
  * results are real (the reference's Mat2Dc always has zero imaginary parts);
  * shape errors raise PlsHipError(INVALID) where the reference only assert()s
    (src/pls.cpp:345-347, :440, :445).
torch is used for device memory and streams only; all arithmetic on the N-sized data is done
by the HIP kernels behind the C-ABI.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib as L

try:  # torch is plumbing (device memory, streams); the host path works without it
    import torch
except Exception:  # pragma: no cover
    torch = None

KERNEL_TYPE1 = L.KERNEL_TYPE1
KERNEL_TYPE2 = L.KERNEL_TYPE2


def _is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


# ---------------------------------------------------------------------------------------------
# column-major device matrices as torch views
# ---------------------------------------------------------------------------------------------

def colmajor_empty(rows: int, cols: int, dtype, device, ld: int | None = None):
    """(rows, cols) view with strides (1, ld): the reference's Eigen column-major layout
    (include/PLS/pls.h:22-23).  ld defaults to rows rounded up to a 16-byte multiple."""
    es = torch.empty((), dtype=dtype).element_size()
    v = 16 // es
    if ld is None:
        ld = max(rows, 1)
        ld += (-ld) % v
    buf = torch.empty((cols, ld), dtype=dtype, device=device)
    return buf[:, :rows].t()


def as_colmajor(x, dtype=None):
    """Zero-copy if x already has strides (1, ld>=rows) and the dtype matches, else a copy."""
    if dtype is None:
        dtype = x.dtype
    if x.dim() == 1:
        x = x[:, None]
    n, k = x.shape
    ok = x.dtype == dtype and (n <= 1 or x.stride(0) == 1) and (k <= 1 or x.stride(1) >= max(n, 1))
    if ok:
        return x
    out = colmajor_empty(n, k, dtype, x.device)
    out.copy_(x)
    return out


def _ld(x) -> int:
    """leading dimension of a column-major view (a single column may report any stride)"""
    n, k = x.shape
    return max(int(x.stride(1)) if k > 1 else 0, int(n), 1)


def _np_f(x, dtype):
    a = np.asarray(x)
    if a.ndim == 1:
        a = a[:, None]
    return np.asfortranarray(a, dtype=dtype)


# ---------------------------------------------------------------------------------------------
# handle
# ---------------------------------------------------------------------------------------------

class Handle:
    """One pls_hip context: a device, a stream, its workspace and (optionally) a reducer."""

    def __init__(self, device: int | None = None, stream: int | None = None):
        self._lib = L.lib()
        if device is None:
            device = torch.cuda.current_device() if (torch is not None and torch.cuda.is_available()) else 0
        if stream is None and torch is not None and torch.cuda.is_available():
            stream = torch.cuda.current_stream(device).cuda_stream
        self.device = int(device)
        h = ctypes.c_void_p()
        L.check(self._lib.pls_hip_create(ctypes.byref(h), self.device, ctypes.c_void_p(stream or 0)))
        self.h = h
        self._keep = []  # objects the C side points at (callbacks, reduce buffers)

    def close(self):
        if getattr(self, "h", None):
            self._lib.pls_hip_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt: int, value: int):
        L.check(self._lib.pls_hip_set_option(self.h, opt, int(value)), self.h)

    def get_option(self, opt: int) -> int:
        v = ctypes.c_int64()
        L.check(self._lib.pls_hip_get_option(self.h, opt, ctypes.byref(v)), self.h)
        return int(v.value)

    def set_stream(self, stream: int | None):
        L.check(self._lib.pls_hip_set_stream(self.h, ctypes.c_void_p(stream or 0)), self.h)

    def synchronize(self):
        L.check(self._lib.pls_hip_synchronize(self.h), self.h)

    def clear_reducer(self):
        """back to a single-rank handle (the reducer installed by pls_amd.distributed.attach_reducer is dropped)"""
        self.synchronize()
        L.check(self._lib.pls_hip_set_reduce_buffer(self.h, None, 0), self.h)
        L.check(self._lib.pls_hip_set_reducer(self.h, L.ALLREDUCE_FN(), None, 0, 1), self.h)

    def timing(self) -> dict:
        t = L.Timing()
        L.check(self._lib.pls_hip_get_timing(self.h, ctypes.byref(t)), self.h)
        return {"fit_ms": t.fit_ms, "fits": int(t.fits),
                "ms": {n: t.fam_ms[i] for i, n in enumerate(L.FAM_NAMES)},
                "launches": {n: int(t.fam_launches[i]) for i, n in enumerate(L.FAM_NAMES)},
                "bytes": {n: int(t.fam_bytes[i]) for i, n in enumerate(L.FAM_NAMES)}}

    # ---- raw steps on device tensors (tests, bench) -----------------------------------------
    def _dt(self, x):
        return L.F64 if x.dtype == torch.float64 else L.F32

    def fit_device(self, X, Y, A: int, method: int = KERNEL_TYPE1, want_B: bool = True, out=None):
        """X (N,K), Y (N,M) column-major CUDA tensors of one dtype.  Enqueues the fit and
        returns dict(W,P,Q,R,T,B) of device tensors (column-major views).  `out`: a dict
        returned by an earlier call with the same shapes, to be overwritten in place."""
        X = as_colmajor(X)
        Y = as_colmajor(Y, X.dtype)
        N, K = X.shape
        M = Y.shape[1]
        dev = X.device
        f64 = torch.float64
        if out is not None:
            W, P, Q, R, T, B = (out[k] for k in "WPQRTB")
            want_B = B is not None
        else:
            W = colmajor_empty(K, A, f64, dev, ld=K); P = colmajor_empty(K, A, f64, dev, ld=K)
            R = colmajor_empty(K, A, f64, dev, ld=K); Q = colmajor_empty(M, A, f64, dev, ld=M)
            # the scores exist for KERNEL_TYPE1 only (reference src/pls.cpp:394,434): None for KERNEL_TYPE2
            T = colmajor_empty(N, A, X.dtype, dev) if method == KERNEL_TYPE1 else None
            B = colmajor_empty(K, M, f64, dev, ld=K) if want_B else None
        if method == KERNEL_TYPE1 and T is None:
            raise L.PlsHipError(L.ERR_INVALID, "KERNEL_TYPE1 needs a T buffer in `out`")
        rc = self._lib.pls_hip_fit(self.h, X.data_ptr(), _ld(X), Y.data_ptr(), _ld(Y), N, K, M, A,
                                   method, self._dt(X), L.MEM_DEVICE, W.data_ptr(), P.data_ptr(),
                                   Q.data_ptr(), R.data_ptr(), T.data_ptr() if T is not None else None,
                                   _ld(T) if T is not None else max(N, 1), B.data_ptr() if want_B else None)
        L.check(rc, self.h)
        self._last_inputs = (X, Y)  # keep alive until the stream has consumed them
        return dict(W=W, P=P, Q=Q, R=R, T=T if method == KERNEL_TYPE1 else None, B=B)

    def fit_host(self, X, Y, A: int, method: int = KERNEL_TYPE1, dtype=np.float64):
        X = _np_f(X, dtype); Y = _np_f(Y, dtype)
        N, K = X.shape
        M = Y.shape[1]
        W = np.zeros((K, A), order="F"); P = np.zeros((K, A), order="F")
        R = np.zeros((K, A), order="F"); Q = np.zeros((M, A), order="F")
        B = np.zeros((K, M), order="F")
        T = np.zeros((N, A), dtype=dtype, order="F")
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = self._lib.pls_hip_fit(self.h, p(X), max(N, 1), p(Y), max(N, 1), N, K, M, A, method,
                                   L.F64 if dtype == np.float64 else L.F32, L.MEM_HOST,
                                   p(W), p(P), p(Q), p(R), p(T), max(N, 1), p(B))
        L.check(rc, self.h)
        return dict(W=W, P=P, Q=Q, R=R, T=T if method == KERNEL_TYPE1 else None, B=B)

    def xb(self, X, Bm):
        """X (N,K) @ Bm (K,C): fitted_values / scores product."""
        if _is_torch(X):
            X = as_colmajor(X)
            Bm = as_colmajor(Bm.to(torch.float64))
            N, K = X.shape
            C = Bm.shape[1]
            out = colmajor_empty(N, C, X.dtype, X.device)
            rc = self._lib.pls_hip_xb(self.h, X.data_ptr(), _ld(X), N, K, Bm.data_ptr(), _ld(Bm), C,
                                      self._dt(X), L.MEM_DEVICE, out.data_ptr(), _ld(out))
            L.check(rc, self.h)
            self._last_inputs = (X, Bm)
            return out
        dtype = np.float32 if np.asarray(X).dtype == np.float32 else np.float64
        X = _np_f(X, dtype); Bm = _np_f(Bm, np.float64)
        N, K = X.shape
        C = Bm.shape[1]
        out = np.zeros((N, C), dtype=dtype, order="F")
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = self._lib.pls_hip_xb(self.h, p(X), max(N, 1), N, K, p(Bm), K, C,
                                  L.F64 if dtype == np.float64 else L.F32, L.MEM_HOST, p(out), max(N, 1))
        L.check(rc, self.h)
        return out

    def coefficients(self, R, Q, c: int):
        if _is_torch(R):
            R = as_colmajor(R); Q = as_colmajor(Q)
            K, A = R.shape
            M = Q.shape[0]
            if _ld(R) != K or _ld(Q) != M:
                R = colmajor_empty(K, A, torch.float64, R.device, ld=K).copy_(R)
                Q = colmajor_empty(M, A, torch.float64, Q.device, ld=M).copy_(Q)
            B = colmajor_empty(K, M, torch.float64, R.device, ld=K)
            rc = self._lib.pls_hip_coefficients(self.h, R.data_ptr(), Q.data_ptr(), K, M, A, c,
                                                L.MEM_DEVICE, B.data_ptr())
            L.check(rc, self.h)
            self._last_inputs = (R, Q)
            return B
        R = _np_f(R, np.float64); Q = _np_f(Q, np.float64)
        K, A = R.shape
        M = Q.shape[0]
        B = np.zeros((K, M), order="F")
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        L.check(self._lib.pls_hip_coefficients(self.h, p(R), p(Q), K, M, A, c, L.MEM_HOST, p(B)), self.h)
        return B

    def xty(self, X, Y):
        X = as_colmajor(X); Y = as_colmajor(Y, X.dtype)
        N, K = X.shape
        M = Y.shape[1]
        out = colmajor_empty(K, M, torch.float64, X.device, ld=K)
        L.check(self._lib.pls_hip_xty(self.h, X.data_ptr(), _ld(X), Y.data_ptr(), _ld(Y), N, K, M,
                                      self._dt(X), out.data_ptr()), self.h)
        self._last_inputs = (X, Y)
        return out

    def deflate(self, src, t, p, dst=None):
        src = as_colmajor(src)
        N, K = src.shape
        if dst is None:
            dst = colmajor_empty(N, K, src.dtype, src.device)
        t = t.to(src.dtype).contiguous()
        p = p.to(torch.float64).contiguous()
        L.check(self._lib.pls_hip_deflate(self.h, src.data_ptr(), _ld(src), dst.data_ptr(), _ld(dst),
                                          N, K, t.data_ptr(), p.data_ptr(), self._dt(src)), self.h)
        self._last_inputs = (src, t, p)
        return dst

    def colwise_z_scores(self, X, n_total: int | None = None, inplace: bool = False):
        """(Z, mean, sd) of a device matrix: colwise_z_scores of the reference (src/pls.cpp:107-111)."""
        X = as_colmajor(X)
        N, K = X.shape
        Z = X if inplace else colmajor_empty(N, K, X.dtype, X.device)
        mean = torch.empty(K, dtype=torch.float64, device=X.device)
        sd = torch.empty(K, dtype=torch.float64, device=X.device)
        L.check(self._lib.pls_hip_colwise_z_scores(self.h, X.data_ptr(), _ld(X), N, n_total or N, K, self._dt(X),
                                                   Z.data_ptr(), _ld(Z), mean.data_ptr(), sd.data_ptr()), self.h)
        self._last_inputs = (X,)
        return Z, mean, sd

    def sse_by_components(self, S, Y, Q):
        """SSE (M x A) for 1..A components from the scores S = X R (one sweep)."""
        S = as_colmajor(S); Y = as_colmajor(Y, S.dtype)
        N, A = S.shape
        M = Y.shape[1]
        Q = as_colmajor(Q.to(torch.float64))
        if _ld(Q) != M:
            Q = colmajor_empty(M, A, torch.float64, Q.device, ld=M).copy_(Q)
        out = colmajor_empty(M, A, torch.float64, S.device, ld=M)
        L.check(self._lib.pls_hip_sse_by_components(self.h, S.data_ptr(), _ld(S), Y.data_ptr(), _ld(Y), N, A, M,
                                                    Q.data_ptr(), self._dt(S), out.data_ptr()), self.h)
        self._last_inputs = (S, Y, Q)
        return out

    def cv_folds(self, X, Y, A: int, test_idx):
        """Residuals of batched cross-validation folds (cv_LOO / cv_LSO of the reference in one launch).
        test_idx: (num_folds, test_size) integer array of held-out rows.  Returns E with shape (M, nobs, A),
        nobs = num_folds * test_size: E[m] is what Residual.errors()[m] holds."""
        idx = np.ascontiguousarray(np.asarray(test_idx, dtype=np.int64))
        if idx.ndim == 1:
            idx = idx[:, None]
        nf, ts = idx.shape
        if _is_torch(X):
            X = as_colmajor(X); Y = as_colmajor(Y, X.dtype)
            N, K = X.shape
            M = Y.shape[1]
            E = torch.empty((M, A, nf * ts), dtype=torch.float64, device=X.device)
            rc = self._lib.pls_hip_cv_folds(self.h, X.data_ptr(), _ld(X), Y.data_ptr(), _ld(Y), N, K, M, A,
                                            idx.ctypes.data_as(ctypes.c_void_p), ts, nf, self._dt(X), L.MEM_DEVICE,
                                            E.data_ptr())
            L.check(rc, self.h)
            return E.permute(0, 2, 1)
        dt = np.float32 if np.asarray(X).dtype == np.float32 else np.float64
        X = _np_f(X, dt); Y = _np_f(Y, dt)
        N, K = X.shape
        M = Y.shape[1]
        E = np.zeros((M, A, nf * ts))
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = self._lib.pls_hip_cv_folds(self.h, p(X), N, p(Y), N, N, K, M, A, p(idx), ts, nf,
                                        L.F64 if dt == np.float64 else L.F32, L.MEM_HOST, p(E))
        L.check(rc, self.h)
        return E.transpose(0, 2, 1)

    def synth_x(self, row0: int, nrows: int, K: int, seed: int, dtype=None, device=None):
        dtype = dtype or torch.float64
        X = colmajor_empty(nrows, K, dtype, device or f"cuda:{self.device}")
        L.check(self._lib.pls_hip_synth_x(self.h, X.data_ptr(), _ld(X), row0, nrows, K, seed,
                                          L.F64 if dtype == torch.float64 else L.F32), self.h)
        return X

    def synth_y(self, row0: int, nrows: int, M: int, seed: int, dtype=None, device=None):
        dtype = dtype or torch.float64
        Y = colmajor_empty(nrows, M, dtype, device or f"cuda:{self.device}")
        L.check(self._lib.pls_hip_synth_y(self.h, Y.data_ptr(), _ld(Y), row0, nrows, M, seed,
                                          L.F64 if dtype == torch.float64 else L.F32), self.h)
        return Y


# ---------------------------------------------------------------------------------------------
# one process, several GPUs (include/pls_hip.h: pls_hip_group) -- what the C++ PLS::Model drives
# ---------------------------------------------------------------------------------------------

class Group:
    """n member handles, one per entry of `devices` (an ordinal may repeat: virtual shards on one GPU), one host
    thread per member inside the library, row-sharded resident matrices, in-process fixed-order all-reduce.
    Host (numpy) matrices in, host results out."""

    def __init__(self, devices):
        self._lib = L.lib()
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        g = ctypes.c_void_p()
        L.check(self._lib.pls_hip_group_create(ctypes.byref(g), len(devices), devs))
        self.g = g
        self.n = len(devices)
        self.devices = [int(d) for d in devices]  # member r lives on device self.devices[r]

    def _check(self, rc):
        if rc != L.OK:
            raw = self._lib.pls_hip_group_last_error(self.g)
            raise L.PlsHipError(rc, raw.decode("utf-8", "replace") if raw else "")

    def close(self):
        if getattr(self, "g", None):
            self._lib.pls_hip_group_destroy(self.g)
            self.g = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt: int, value: int):
        self._check(self._lib.pls_hip_group_set_option(self.g, opt, int(value)))

    @property
    def exchange(self) -> str:
        """'device': the members push their partial sums into each other's inboxes (no host in a collective);
        'host': the host-synchronised exchange (members sharing a GPU, or a single member)"""
        return "device" if self._lib.pls_hip_group_exchange(self.g) else "host"

    def upload(self, a, dtype=np.float64):
        a = _np_f(a, dtype)
        m = ctypes.c_void_p()
        self._check(self._lib.pls_hip_group_upload(self.g, a.ctypes.data_as(ctypes.c_void_p), max(a.shape[0], 1), a.shape[0],
                                                   a.shape[1], L.F64 if dtype == np.float64 else L.F32, ctypes.byref(m)))
        return m

    def upload_xy(self, X, Y, dtype=np.float64):
        """X and Y of one data set; X^T X and X^T Y are accumulated on the matrix cores while X streams in and kept
        with the pair for fits under ALGO_AUTO / ALGO_GRAM / KERNEL_TYPE2"""
        X = _np_f(X, dtype); Y = _np_f(Y, dtype)
        mx, my = ctypes.c_void_p(), ctypes.c_void_p()
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self._check(self._lib.pls_hip_group_upload_xy(self.g, p(X), max(X.shape[0], 1), p(Y), max(Y.shape[0], 1), X.shape[0],
                                                      X.shape[1], Y.shape[1], L.F64 if dtype == np.float64 else L.F32,
                                                      ctypes.byref(mx), ctypes.byref(my)))
        return mx, my

    def alloc(self, N: int, K: int, dtype=np.float64):
        m = ctypes.c_void_p()
        self._check(self._lib.pls_hip_group_alloc(self.g, N, K, L.F64 if dtype == np.float64 else L.F32, ctypes.byref(m)))
        return m

    def member_handle(self, r: int):
        h = ctypes.c_void_p()
        L.check(self._lib.pls_hip_group_handle(self.g, r, ctypes.byref(h)))
        return h

    def block(self, m, r: int):
        """(device pointer, ld, row0, nrows) of member r's block of a resident matrix"""
        ptr, ld, r0, nr = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        L.check(self._lib.pls_hip_matrix_block(m, r, ctypes.byref(ptr), ctypes.byref(ld), ctypes.byref(r0), ctypes.byref(nr)))
        return ptr, int(ld.value), int(r0.value), int(nr.value)

    def synth(self, N: int, cols: int, seed: int, which: str = "x", dtype=np.float64):
        """resident N x cols matrix of the synthetic generator (DESIGN.md "Synthetic inputs"), every member generating
        its own rows on its device -- no host copy (config 5 at its own size: 137 GB)"""
        m = self.alloc(N, cols, dtype)
        fn = self._lib.pls_hip_synth_x if which == "x" else self._lib.pls_hip_synth_y
        for r in range(self.n):
            h = self.member_handle(r)
            ptr, ld, r0, nr = self.block(m, r)
            L.check(fn(h, ptr, ld, r0, nr, cols, seed, L.F64 if dtype == np.float64 else L.F32), h)
        for r in range(self.n):
            L.check(self._lib.pls_hip_synchronize(self.member_handle(r)))
        return m

    def gram(self, Am, Bm):
        """A^T B (cols x cols, numpy) of two resident matrices with the same row partition: each member's block product
        on its device (pls_hip_xty), summed on the host.  Size-independent checks on matrices that never leave the GPUs."""
        ka, kb = self.shape(Am)[1], self.shape(Bm)[1]
        import torch
        total = np.zeros((ka, kb))
        for r in range(self.n):
            h = self.member_handle(r)
            pa, lda, _, nr = self.block(Am, r)
            pb, ldb, _, _ = self.block(Bm, r)
            if nr == 0:
                continue
            out = torch.empty((kb, ka), dtype=torch.float64, device=f"cuda:{self.devices[r]}")  # column-major ka x kb, on the member's GPU
            L.check(self._lib.pls_hip_xty(h, pa, lda, pb, ldb, nr, ka, kb, L.F64 if self.shape(Am)[2] == np.float64 else L.F32,
                                          out.data_ptr()), h)
            L.check(self._lib.pls_hip_synchronize(h), h)
            total += out.cpu().numpy().T
        return total

    def shape(self, m):
        n, k, dt = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        L.check(self._lib.pls_hip_matrix_shape(m, ctypes.byref(n), ctypes.byref(k), ctypes.byref(dt)))
        return int(n.value), int(k.value), (np.float64 if dt.value == L.F64 else np.float32)

    def blocks(self, m):
        """[(row0, nrows)] of the members"""
        out = []
        for r in range(self.n):
            r0, nr = ctypes.c_int64(), ctypes.c_int64()
            L.check(self._lib.pls_hip_matrix_block(m, r, None, None, ctypes.byref(r0), ctypes.byref(nr)))
            out.append((int(r0.value), int(nr.value)))
        return out

    def download(self, m, col0: int = 0, ncols: int | None = None):
        N, K, dt = self.shape(m)
        ncols = K - col0 if ncols is None else ncols
        out = np.zeros((N, ncols), dtype=dt, order="F")
        self._check(self._lib.pls_hip_group_download(self.g, m, col0, ncols, out.ctypes.data_as(ctypes.c_void_p), N))
        return out

    def free(self, m):
        self._check(self._lib.pls_hip_group_free(self.g, m))

    def fit(self, X, Y, A: int, method: int = KERNEL_TYPE1):
        """X, Y: resident matrices.  Returns dict(W,P,Q,R,B numpy; T resident matrix or None)."""
        N, K, dt = self.shape(X)
        M = self.shape(Y)[1]
        W = np.zeros((K, A), order="F"); P = np.zeros((K, A), order="F"); R = np.zeros((K, A), order="F")
        Q = np.zeros((M, A), order="F"); B = np.zeros((K, M), order="F")
        T = self.alloc(N, A, dt) if method == KERNEL_TYPE1 else None
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self._check(self._lib.pls_hip_group_fit(self.g, X, Y, A, method, p(W), p(P), p(Q), p(R), T, p(B)))
        return dict(W=W, P=P, Q=Q, R=R, T=T, B=B)

    def xb(self, X, Bm):
        N, K, dt = self.shape(X)
        Bm = _np_f(Bm, np.float64)
        out = self.alloc(N, Bm.shape[1], dt)
        self._check(self._lib.pls_hip_group_xb(self.g, X, Bm.ctypes.data_as(ctypes.c_void_p), K, Bm.shape[1], out))
        return out

    def model_sse(self, X, Y, R, Q):
        R = _np_f(R, np.float64); Q = _np_f(Q, np.float64)
        M, A = Q.shape
        sse = np.zeros((M, A), order="F")
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self._check(self._lib.pls_hip_group_model_sse(self.g, X, Y, A, p(R), p(Q), p(sse)))
        return sse

    def cv_folds(self, X, Y, A: int, test_idx):
        idx = np.ascontiguousarray(np.asarray(test_idx, dtype=np.int64))
        if idx.ndim == 1:
            idx = idx[:, None]
        nf, ts = idx.shape
        M = self.shape(Y)[1]
        E = np.zeros((M, A, nf * ts))
        self._check(self._lib.pls_hip_group_cv_folds(self.g, X, Y, A, idx.ctypes.data_as(ctypes.c_void_p), ts, nf,
                                                     E.ctypes.data_as(ctypes.c_void_p)))
        return E.transpose(0, 2, 1)


# ---------------------------------------------------------------------------------------------
# PLS::Model
# ---------------------------------------------------------------------------------------------

class Model:
    """Mirror of `PLS::Model` (reference include/PLS/pls.h:184-266).

    Model(X, Y, algorithm=KERNEL_TYPE1, max_components=None) fits immediately, like the
    reference's data-taking constructors (src/pls.cpp:340-359): max_components defaults to
    X.cols().  X: N x K, Y: N x M.  torch CUDA tensors stay on the device; numpy arrays go
    through the library's host path.
    """

    def __init__(self, X, Y, algorithm: int = KERNEL_TYPE1, max_components: int | None = None, *,
                 handle: Handle | None = None):
        self.handle = handle or Handle()
        self._on_device = _is_torch(X)
        K = X.shape[1]
        self.A = int(K if max_components is None else max_components)  # src/pls.cpp:356-359
        self.method = algorithm
        self.W = self.P = self.R = self.Q = self.T = None
        self._X, self._Y = X, Y  # the reference keeps _X, _Y for its cross-validation methods (pls.h:250)
        self.plsr(X, Y, algorithm)

    # void plsr(const Mat2D&, const Mat2D&, const METHOD&)  include/PLS/pls.h:199
    def plsr(self, X, Y, algorithm: int = KERNEL_TYPE1):
        self.method = algorithm
        if _is_torch(X):
            out = self.handle.fit_device(X, Y, self.A, algorithm, want_B=False)
        else:
            dt = np.float32 if np.asarray(X).dtype == np.float32 else np.float64
            out = self.handle.fit_host(X, Y, self.A, algorithm, dtype=dt)
        self.W, self.P, self.Q, self.R, self.T = (out[k] for k in "WPQRT")

    def _comp(self, comp):
        comp = self.A if comp is None else int(comp)
        if not (0 <= comp <= self.A):  # assert (A >= comp): src/pls.cpp:440, :445
            raise L.PlsHipError(L.ERR_INVALID, f"comp={comp} exceeds the fitted A={self.A}")
        return comp

    # const Mat2Dc scores(const Mat2D& X_new, size_t comp)   src/pls.cpp:439-442
    def scores(self, X_new, comp: int | None = None):
        c = self._comp(comp)
        return self.handle.xb(X_new, self.R[:, :c])

    # declared but never defined by the reference (include/PLS/pls.h:207-211); the natural
    # definitions, SURVEY.md section 8(b)
    def loadingsX(self, comp: int | None = None):
        return self.P[:, :self._comp(comp)]

    def loadingsY(self, comp: int | None = None):
        return self.Q[:, :self._comp(comp)]

    # const Mat2Dc coefficients(size_t comp)   src/pls.cpp:444-447
    def coefficients(self, comp: int | None = None):
        return self.handle.coefficients(self.R, self.Q, self._comp(comp))

    # const Mat2D fitted_values(const Mat2D& X, size_t comp)   src/pls.cpp:449-451
    def fitted_values(self, X, comp: int | None = None):
        return self.handle.xb(X, self.coefficients(comp))

    # residuals / SSE / explained_variance: src/pls.cpp:453-467 (thin host arithmetic on N x M)
    def residuals(self, X, Y, comp: int | None = None):
        fv = self.fitted_values(X, comp)
        if _is_torch(Y):
            Y2 = Y if Y.dim() == 2 else Y[:, None]
            return Y2.to(fv.dtype) - fv
        return _np_f(Y, fv.dtype) - fv

    def SSE(self, X, Y, comp: int | None = None):
        r = self.residuals(X, Y, comp)
        return (r * r).sum(0)

    # ---- cross-validation (reference include/PLS/pls.h:235-241, src/pls.cpp:469-549) -------------------
    # Each returns the residual tensor E of shape (M, nobs, A): E[m] is Residual.errors()[m].
    def cv_LOO(self):
        """leave-one-out: fold i refits on all rows but i (all folds in one batched device call)"""
        n = self._X.shape[0]
        return self.handle.cv_folds(self._X, self._Y, self.A, np.arange(n)[:, None])

    def cv_LSO(self, test_fraction: float, num_trials: int, rng=None):
        """leave-some-out: num_trials random splits with round(test_fraction*N) held-out rows each.
        The reference draws its splits from std::shuffle on a std::mt19937 (src/pls.cpp:218-227);
        here they come from a numpy Generator (pass one for reproducibility)."""
        n = self._X.shape[0]
        ts = int(test_fraction * n + 0.5)
        if ts == 0 or ts == n:
            raise L.PlsHipError(L.ERR_INVALID, "empty train or test split")
        rng = np.random.default_rng() if rng is None else rng
        idx = np.stack([rng.permutation(n)[:ts] for _ in range(num_trials)])
        return self.handle.cv_folds(self._X, self._Y, self.A, idx)

    def cv_NEW_DATA(self, X_new, Y_new):
        """residuals on data outside the fit for 1..A components (src/pls.cpp:494-509); no refit"""
        S = self.scores(X_new)
        if _is_torch(S):
            Y2 = (Y_new if Y_new.dim() == 2 else Y_new[:, None]).to(torch.float64)
            fit = torch.cumsum(S.to(torch.float64)[:, :, None] * self.Q.t()[None, :, :], dim=1)  # (N, A, M)
            return (Y2[:, None, :] - fit).permute(2, 0, 1)
        Y2 = _np_f(Y_new, np.float64)
        fit = np.cumsum(np.asarray(S, dtype=np.float64)[:, :, None] * np.asarray(self.Q).T[None, :, :], axis=1)
        return (Y2[:, None, :] - fit).transpose(2, 0, 1)

    def explained_variance_by_components(self, X, Y):
        """(EV, SSE), each M x A: what print_explained_variance (src/pls.cpp:551-562) reports for
        ncomp = 1..A, from one X*R pass and one sweep over the scores instead of A X*B passes.
        Device tensors only."""
        X = as_colmajor(X)
        Y2 = as_colmajor(Y if Y.dim() == 2 else Y[:, None], X.dtype)
        N, K = X.shape
        M = Y2.shape[1]
        sse = colmajor_empty(M, self.A, torch.float64, X.device, ld=M)
        L.check(L.lib().pls_hip_model_sse(self.handle.h, X.data_ptr(), _ld(X), Y2.data_ptr(), _ld(Y2), N, K, M,
                                          self.A, self.R.data_ptr(), self.Q.data_ptr(), self.handle._dt(X),
                                          L.MEM_DEVICE, sse.data_ptr()), self.handle.h)
        Yd = Y2.to(torch.float64)
        sst = ((Yd - Yd.mean(0, keepdim=True)) ** 2).sum(0) if Yd.shape[0] >= 2 else torch.zeros(Yd.shape[1], device=Yd.device)
        return 1.0 - sse / sst[:, None], sse

    def explained_variance(self, X, Y, comp: int | None = None):
        sse = self.SSE(X, Y, comp)
        Y2 = Y if Y.ndim == 2 else Y[:, None]
        if _is_torch(Y2):
            Yd = Y2.to(torch.float64)
            sst = ((Yd - Yd.mean(0, keepdim=True)) ** 2).sum(0) if Yd.shape[0] >= 2 else torch.zeros_like(sse)
            return 1.0 - sse.to(torch.float64) / sst
        Yd = np.asarray(Y2, dtype=np.float64)
        sst = ((Yd - Yd.mean(0)) ** 2).sum(0) if Yd.shape[0] >= 2 else np.zeros(Yd.shape[1])  # SST: :69-73
        return 1.0 - np.asarray(sse, dtype=np.float64) / sst
