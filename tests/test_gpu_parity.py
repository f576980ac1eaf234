"""GPU parity tests: the HIP path (through the C-ABI) against the oracle and the golden fixtures.

Tolerances (fp64): coefficients B within 1e-10 relative Frobenius error of the oracle (the north
star's bar); W,P,Q,R,T compared per component modulo sign (the reference leaves the eigenvector
sign open) within 1e-9.  Integer/bit-exact checks: the synthetic generator.
"""
import glob
import subprocess
import sys
import os

import numpy as np
import pytest

from conftest import handle_with_env, DATA, GOLDEN, ROOT

pytestmark = pytest.mark.gpu

TOL_B = 1e-10
TOL_COL = 1e-9


def _torch():
    import torch
    return torch


def to_dev(a, dtype=None):
    torch = _torch()
    import pls_amd
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return pls_amd.as_colmajor(t if dtype is None else t.to(dtype))


def check_against(po, out, ref, Bref, Tref=None, tol_b=TOL_B, tol_col=TOL_COL, col_err=None, tol_inv=1e-8):
    """B: relative Frobenius error <= tol_b (sign-invariant, well conditioned).
    W,P,R,Q,T: per component, modulo sign, <= max(tol_col, 20 x col_err[a]) where col_err is the
    disagreement between independent fp64 CPU implementations on that component (noise
    components are ill-conditioned in direction, not in B; see tests/golden/make_golden.py)."""
    got = {k: out[k].cpu().numpy().astype(np.float64) for k in "WPQRB"}
    assert np.isfinite(got["B"]).all()
    assert po.rel_fro(got["B"], Bref) < tol_b
    refd = {k: np.asarray(ref[k]) for k in "WPQR"}
    refd["T"] = None if Tref is None else np.asarray(Tref)
    got["T"] = None if Tref is None else out["T"].cpu().numpy().astype(np.float64)
    err = po.column_errors(refd, got)
    A = got["W"].shape[1]
    lim = np.full(A, tol_col) if col_err is None else np.maximum(tol_col, 20.0 * np.asarray(col_err))
    assert (err <= lim).all(), f"column errors {err} exceed {lim}"
    # invariants of the algorithm: unit-norm w, P^T R = I
    assert np.allclose((got["W"] ** 2).sum(0), 1.0, atol=1e-12)
    assert np.allclose(got["P"].T @ got["R"], np.eye(A), atol=tol_inv)


def oracle_ref(oracle, po, Xh, Yh, A):
    """oracle fit + coefficient matrix + per-component conditioning estimate (kernel form vs the
    independently written NIPALS-deflation form of the oracle)."""
    ref = oracle.plsr(Xh, Yh, A)
    alt = oracle.plsr(Xh, Yh, A, nipals=True)
    return ref, oracle.coefficients(ref["R"], ref["Q"]), po.column_errors(ref, alt)


def golden_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "s_*.npz")))


@pytest.fixture(params=[(0, 0), (0, 1), (1, 0), (1, 1), (2, 1), (3, 1)],
                ids=["kernel-unfused", "kernel-fused", "nipals-unfused", "nipals-fused", "gram", "auto"])
def mode(request, handle):
    import pls_amd
    algo, fuse = request.param
    handle.set_option(pls_amd.OPT_ALGO, algo)
    handle.set_option(pls_amd.OPT_FUSE, fuse)
    yield request.param
    handle.set_option(pls_amd.OPT_ALGO, 0)
    handle.set_option(pls_amd.OPT_FUSE, 1)


# ------------------------------------------------------------------------------------------
# reference example data (BASELINE configs 1 and 2)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,fx,fy", [("toy_A2", "toyX.csv", "toyY.csv"), ("nir_A10", "nir.csv", "octane.csv")])
def test_reference_csv_golden(handle, oracle, po, mode, name, fx, fy):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    A = int(g["A"])
    X = oracle.z_scores(po.read_csv(os.path.join(DATA, fx)))
    Y = oracle.z_scores(po.read_csv(os.path.join(DATA, fy)))
    out = handle.fit_device(to_dev(X), to_dev(Y), A)
    handle.synchronize()
    check_against(po, out, g, g["B"], g["T"], col_err=g["col_err"])
    assert np.allclose((out["T"].cpu().numpy() ** 2).sum(0), g["tt"], rtol=1e-9)


def test_reference_csv_model_api(handle, oracle, po):
    """the Python mirror of PLS::Model on the README smoke test (toy, A=2): fit, coefficients,
    fitted_values, scores, explained variance / SSE (src/pls.cpp:439-467)."""
    import pls_amd
    g = np.load(os.path.join(GOLDEN, "toy_A2.npz"))
    X = oracle.z_scores(po.read_csv(os.path.join(DATA, "toyX.csv")))
    Y = oracle.z_scores(po.read_csv(os.path.join(DATA, "toyY.csv")))
    Xd, Yd = to_dev(X), to_dev(Y)
    m = pls_amd.Model(Xd, Yd, pls_amd.KERNEL_TYPE1, 2, handle=handle)
    assert po.rel_fro(m.coefficients().cpu().numpy(), g["B"]) < TOL_B
    for c in (1, 2):
        ev = m.explained_variance(Xd, Yd, c).cpu().numpy()
        sse = m.SSE(Xd, Yd, c).cpu().numpy()
        assert np.allclose(ev, g["explained_variance"][c - 1], rtol=1e-9, atol=1e-12)
        assert np.allclose(sse, g["SSE"][c - 1], rtol=1e-9)
    s = po.sign_align(g["W"], m.W.cpu().numpy())
    assert po.rel_fro(m.scores(Xd).cpu().numpy() * s, g["T"]) < TOL_COL
    assert po.rel_fro(m.fitted_values(Xd).cpu().numpy(), X @ g["B"]) < TOL_B
    with pytest.raises(pls_amd.PlsHipError):
        m.coefficients(3)  # comp > A: the reference asserts (src/pls.cpp:445)
    # host-memory path (numpy in, numpy out): what the C++ Model uses
    mh = pls_amd.Model(X, Y, pls_amd.KERNEL_TYPE1, 2, handle=handle)
    assert po.rel_fro(mh.coefficients(), g["B"]) < TOL_B
    assert po.rel_fro(mh.fitted_values(X), X @ g["B"]) < TOL_B


# ------------------------------------------------------------------------------------------
# seeded synthetic shapes: ragged N and K, m in {1,2,4,8}
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_cases())
def test_synthetic_golden(handle, po, oracle, mode, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    N, K, M, A, seed = (int(g[k]) for k in ("N", "K", "M", "A", "seed"))
    X = handle.synth_x(0, N, K, seed)
    Y = handle.synth_y(0, N, M, seed)
    # the device generator is bit-identical to the host one (integer/dyadic arithmetic)
    assert float(np.abs(X.cpu().numpy()).sum()) == float(g["x_checksum"])
    assert float(np.abs(Y.cpu().numpy()).sum()) == float(g["y_checksum"])
    out = handle.fit_device(X, Y, A)
    handle.synchronize()
    check_against(po, out, g, g["B"], g["T"] if "T" in g.files else None, col_err=g["col_err"])
    assert np.allclose((out["T"].cpu().numpy() ** 2).sum(0), g["tt"], rtol=1e-9)


def test_synth_bit_exact(handle, oracle, po):
    for (r0, n, K, M) in ((0, 9, 7, 2), (12345, 1000, 33, 8), (1 << 20, 513, 64, 1)):
        X = handle.synth_x(r0, n, K, 0x504C5301).cpu().numpy()
        Y = handle.synth_y(r0, n, M, 0x504C5301).cpu().numpy()
        assert np.array_equal(X, oracle.synth_x(r0, n, K))
        assert np.array_equal(Y, oracle.synth_y(r0, n, M))
        assert np.array_equal(X, po.synth_x(r0, n, K))
    # fp32 storage = the fp64 value rounded once
    torch = _torch()
    X32 = handle.synth_x(7, 300, 20, 0x504C5301, dtype=torch.float32).cpu().numpy()
    assert np.array_equal(X32, oracle.synth_x(7, 300, 20).astype(np.float32))


# ------------------------------------------------------------------------------------------
# single steps of the path
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,K,M", [(9, 7, 2), (1000, 64, 4), (4097, 513, 8), (2048, 512, 1), (777, 33, 3)])
def test_steps_xty_xb_deflate(handle, oracle, po, N, K, M):
    torch = _torch()
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    X, Y = to_dev(Xh), to_dev(Yh)
    XY = handle.xty(X, Y); handle.synchronize()
    assert po.rel_fro(XY.cpu().numpy(), oracle.xty(Xh, Yh)) < 1e-13
    Bm = np.asfortranarray(np.random.default_rng(1).standard_normal((K, M)))
    out = handle.xb(X, to_dev(Bm)); handle.synchronize()
    assert po.rel_fro(out.cpu().numpy(), oracle.xb(Xh, Bm)) < 1e-13
    t = Yh[:, 0].copy(); p = Bm[:, 0].copy()
    D = handle.deflate(X, torch.from_numpy(t).cuda(), torch.from_numpy(p).cuda()); handle.synchronize()
    assert po.rel_fro(D.cpu().numpy(), Xh - np.outer(t, p)) < 1e-15
    # unaligned leading dimension / odd base pointer take the narrow (8-byte) path
    big = torch.empty((K, N + 3), dtype=torch.float64, device="cuda")
    Xo = big[:, 1:N + 1].t()
    Xo.copy_(X)
    assert po.rel_fro(handle.xty(Xo, Y).cpu().numpy(), oracle.xty(Xh, Yh)) < 1e-13
    assert po.rel_fro(handle.xb(Xo, to_dev(Bm)).cpu().numpy(), oracle.xb(Xh, Bm)) < 1e-13


def test_unaligned_fit(handle, oracle, po, mode):
    torch = _torch()
    N, K, M, A = 1001, 37, 2, 6
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    big = torch.empty((K, N + 3), dtype=torch.float64, device="cuda")
    X = big[:, 1:N + 1].t()
    X.copy_(torch.from_numpy(Xh).cuda())
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    out = handle.fit_device(X, to_dev(Yh), A); handle.synchronize()
    check_against(po, out, ref, Bref, ref["T"], col_err=cerr)


def test_x_not_modified_and_deterministic(handle, oracle, po, mode):
    N, K, M, A = 3000, 96, 3, 7
    X = handle.synth_x(0, N, K, 1234); Y = handle.synth_y(0, N, M, 1234)
    X0 = X.clone()
    a = handle.fit_device(X, Y, A); handle.synchronize()
    b = handle.fit_device(X, Y, A); handle.synchronize()
    assert _torch().equal(X, X0), "the caller's X must never be written (const at the API, pls.h:188)"
    for k in "WPQRTB":  # fixed-order reductions: bit-identical run to run
        assert _torch().equal(a[k], b[k]), k


@pytest.mark.parametrize("N,K,A,dt,algo", [(32 * 300, 512, 9, "f64", 1), (32 * 300 + 1, 200, 6, "f64", 1), (64 * 90, 1000, 5, "f32", 1),
                                           (32 * 64, 100, 4, "f64", 0), (16 * 300, 1024, 4, "f64", 1)])
def test_one_response_update_routes_bit_identical(N, K, A, dt, algo):
    """PLS_HIP_TAIL = 0 / 1 / 2: partial rows summed by reduce_partials_kernel, in the tail of the pass, and with the one-response
    component update as the last act of that tail (one launch per component).  1 and 2 run the same launches with the same
    sums in the same order and share one piece of update arithmetic (update_m1.hpp): W, P, Q, R, T, B equal bit for bit.
    Without the tail (0) a short read-only pass runs two workgroups per CU instead of one -- other partial rows, the same
    sums to rounding."""
    code = '''
import sys, hashlib
sys.path.insert(0, %r)
import torch, pls_amd
h = pls_amd.Handle()
h.set_option(pls_amd.OPT_ALGO, %d)
dt = torch.float64 if %r == "f64" else torch.float32
X = h.synth_x(0, %d, %d, 77, dtype=dt); Y = h.synth_y(0, %d, 1, 77, dtype=dt)
out = h.fit_device(X, Y, %d); h.synchronize()
print("DIGEST", hashlib.sha256(b"".join(out[k].cpu().numpy().tobytes() for k in "WPQRTB")).hexdigest())
import numpy as np
np.save(sys.argv[1], out["B"].cpu().numpy().astype(np.float64))
''' % (ROOT, algo, dt, N, K, N, A)
    import tempfile
    digests, Bs = [], []
    with tempfile.TemporaryDirectory() as td:
        for tail in ("0", "1", "2"):
            f = os.path.join(td, f"b{tail}.npy")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, PLS_HIP_TAIL=tail), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            digests.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0])
            Bs.append(np.load(f))
    assert digests[1] == digests[2], digests
    assert np.linalg.norm(Bs[0] - Bs[1]) <= (1e-12 if dt == "f64" else 1e-5) * np.linalg.norm(Bs[1])


@pytest.mark.parametrize("N,K,M,A,dt", [(3000, 96, 3, 7, "f64"), (4098, 513, 1, 5, "f64"), (32 * 700 + 2, 40, 2, 4, "f64"),
                                        (5000, 200, 2, 6, "f32"), (64 * 33 + 4, 1024, 1, 3, "f32"),
                                        (64 * 33 + 4, 1500, 2, 4, "f32"), (32 * 40 + 2, 1100, 1, 4, "f64")])
def test_work_layouts_bit_identical(handle, N, K, M, A, dt):
    """The NIPALS work buffer (the deflated copy of X) is row-tile-major by default and column-major with
    OPT_WORK_LAYOUT = 0: a storage choice only -- for K <= 512 every output must agree bit for bit (ragged last
    tile, K not a multiple of the 32 column groups, both storage types).  Wider matrices use shorter tiles (or, beyond
    1024 columns, different kernels) on the tiled copy: agreement to rounding there."""
    import pls_amd
    torch = _torch()
    dtype = torch.float64 if dt == "f64" else torch.float32
    X = handle.synth_x(0, N, K, 99, dtype=dtype); Y = handle.synth_y(0, N, M, 99, dtype=dtype)
    handle.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_NIPALS)
    try:
        outs = []
        for layout in (1, 0):
            handle.set_option(pls_amd.OPT_WORK_LAYOUT, layout)
            o = handle.fit_device(X, Y, A); handle.synchronize()
            outs.append({k: v.clone() for k, v in o.items()})
        # bit-identical only while both layouts run the same tile shape: beyond 16 columns per lane (K > 512) the
        # tiled copy uses shorter tiles (different summation grouping), beyond 1024 different kernels
        wide = K > 32 * 16
        for k in "WPQRTB":
            assert torch.isfinite(outs[0][k]).all(), k
            if not wide:
                assert torch.equal(outs[0][k], outs[1][k]), k
            else:
                a, b = outs[0][k].double(), outs[1][k].double()
                assert float((a - b).norm() / b.norm()) < (1e-11 if dt == "f64" else 2e-6), k
    finally:
        handle.set_option(pls_amd.OPT_WORK_LAYOUT, 1)
        handle.set_option(pls_amd.OPT_ALGO, 0)


@pytest.mark.parametrize("name,algo,method,N,K,M,A,dt", [
    ("nipals", 1, 0, 65536, 512, 1, 5, "f64"), ("gram", 2, 0, 65536, 512, 1, 6, "f64"), ("type2-f32", 0, 1, 65536, 384, 3, 6, "f32"),
    ("nipals-wide-f32", 1, 0, 8192, 4096, 8, 5, "f32"), ("kernel-wide", 0, 0, 8192, 2048, 2, 5, "f64"),
    ("nipals-mid", 1, 0, 32768, 1024, 4, 5, "f64")])
def test_race_screen_repeated_fits(handle, name, algo, method, N, K, M, A, dt):
    """Every kernel is deterministic (fixed-order reductions, no atomics on data): a bit that differs between repeated
    fits of the same inputs is a synchronisation bug -- e.g. an LDS-DMA slab of the SYRK read before it landed, a tile
    of the working copy read before the previous pass finished.  (tools/determinism_soak.py runs this 150 times.)"""
    import pls_amd
    torch = _torch()
    dtype = torch.float64 if dt == "f64" else torch.float32
    handle.set_option(pls_amd.OPT_ALGO, algo)
    try:
        X = handle.synth_x(0, N, K, 7, dtype=dtype); Y = handle.synth_y(0, N, M, 7, dtype=dtype)
        ref = {k: v.clone() for k, v in handle.fit_device(X, Y, A, method=method).items() if v is not None}
        handle.synchronize()
        for _ in range(12):
            out = handle.fit_device(X, Y, A, method=method); handle.synchronize()
            for k, v in ref.items():
                if k == "T" and method == 1:
                    continue
                assert torch.equal(out[k], v), (name, k)
    finally:
        handle.set_option(pls_amd.OPT_ALGO, 0)


@pytest.mark.parametrize("N,K,M,A,dt", [(3000, 96, 3, 9, "f64"), (4098, 512, 1, 7, "f64"), (32 * 300 + 2, 40, 2, 5, "f64"),
                                        (5000, 200, 2, 8, "f32"), (2048, 130, 8, 2, "f64")])
@pytest.mark.parametrize("D", [2, 3, 4])
def test_deferred_write_back_matches_explicit_deflation(handle, oracle, po, N, K, M, A, dt, D):
    """OPT_DEFER = D: the deflated matrix is written back every D-th component only and the pending rank-1 updates
    are re-applied in registers with the explicit plan's own roundings.  Same tiles, same bits in the registers:
    W, P, Q, R, T, B must agree with the explicit plan (D = 1) to the rounding of the partial sums, and with the oracle."""
    import pls_amd
    torch = _torch()
    dtype = torch.float64 if dt == "f64" else torch.float32
    X = handle.synth_x(0, N, K, 123, dtype=dtype); Y = handle.synth_y(0, N, M, 123, dtype=dtype)
    handle.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_NIPALS)
    try:
        handle.set_option(pls_amd.OPT_DEFER, 1)
        ref = {k: v.clone() for k, v in handle.fit_device(X, Y, A).items()}; handle.synchronize()
        handle.set_option(pls_amd.OPT_DEFER, D)
        out = handle.fit_device(X, Y, A); handle.synchronize()
        for k in "WPQRTB":
            a, b = out[k].double(), ref[k].double()
            assert torch.isfinite(a).all(), k
            assert float((a - b).norm() / b.norm()) < (1e-11 if dt == "f64" else 5e-6), k
        if dt == "f64":
            Xh = X.cpu().numpy(); Yh = Y.cpu().numpy()
            r, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
            check_against(po, out, r, Bref, r["T"], col_err=cerr)
    finally:
        handle.set_option(pls_amd.OPT_DEFER, 1)
        handle.set_option(pls_amd.OPT_ALGO, 0)


def test_t_orthogonal_full_rank_components(handle, po):
    """A = K components on a small matrix: scores mutually orthogonal, P^T R = I, and the
    regression reproduces least squares (B_A=K == lstsq) -- a size-independent property."""
    N, K, M = 500, 12, 2
    X = handle.synth_x(0, N, K, 99); Y = handle.synth_y(0, N, M, 99)
    out = handle.fit_device(X, Y, K); handle.synchronize()
    T = out["T"].cpu().numpy()
    G = T.T @ T
    off = G - np.diag(np.diag(G))
    assert np.abs(off).max() < 1e-9 * np.diag(G).max()
    Bls = np.linalg.lstsq(X.cpu().numpy(), Y.cpu().numpy(), rcond=None)[0]
    assert po.rel_fro(out["B"].cpu().numpy(), Bls) < 1e-8


def test_power_iteration_close_eigenvalues(handle, oracle, po):
    """m > 1 direction: two response columns of nearly equal weight (eigenvalues of S^T S close)."""
    N, K, A = 2000, 24, 5
    Xh = oracle.synth_x(0, N, K)
    Yh = oracle.synth_y(0, N, 4)
    Yh[:, 1] = Yh[:, 1] * 2.0   # undo the 2^-j scaling so that columns 0 and 1 compete
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    out = handle.fit_device(to_dev(Xh), to_dev(Yh), A); handle.synchronize()
    check_against(po, out, ref, Bref, ref["T"], col_err=cerr)


@pytest.mark.parametrize("N,K,M,A", [(9, 7, 2, 4), (1000, 64, 1, 12), (777, 33, 3, 8), (2048, 130, 8, 6),
                                     # the SYRK's diagonal blocks in PAIRS (M <= 4, two or more column blocks): a pair + a single block
                                     # (K = 384), a ragged second / third block, row counts that are no multiple of the 16-row slab,
                                     # 2 and 4 responses on board, eight waves of workgroups (K >= 1024)
                                     (4104, 384, 1, 8), (3000, 200, 2, 6), (2050, 300, 4, 5), (1040, 256, 1, 7), (3000, 1100, 1, 4),
                                     (33000, 512, 1, 6)])
def test_kernel_type2(handle, oracle, po, N, K, M, A):
    """METHOD::KERNEL_TYPE2 (src/pls.cpp:398,422-425): XX = X^T X once, no pass over X in the loop, T
    not computed.  Same W,P,Q,R,B as KERNEL_TYPE1 up to rounding."""
    import pls_amd
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    ref = oracle.plsr(Xh, Yh, A, method=1)
    _, _, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    out = handle.fit_device(to_dev(Xh), to_dev(Yh), A, method=pls_amd.KERNEL_TYPE2); handle.synchronize()
    check_against(po, out, ref, oracle.coefficients(ref["R"], ref["Q"]), None, col_err=cerr)


@pytest.mark.parametrize("N,K,M,A", [(1500, 70, 2, 5), (5004, 640, 3, 5), (4096, 256, 1, 6)])
def test_kernel_type2_fp32_storage(handle, oracle, po, N, K, M, A):
    """KERNEL_TYPE2 on fp32 storage: fp32 panels through LDS, converted to fp64 at the MFMA operand read (single diagonal blocks,
    pairs with 3 responses on board over five column blocks, pairs at one response)."""
    import pls_amd
    torch = _torch()
    X = handle.synth_x(0, N, K, 21, dtype=torch.float32); Y = handle.synth_y(0, N, M, 21, dtype=torch.float32)
    Xh = X.cpu().numpy().astype(np.float64); Yh = Y.cpu().numpy().astype(np.float64)
    ref = oracle.plsr(Xh, Yh, A, method=1)
    out = handle.fit_device(X, Y, A, method=pls_amd.KERNEL_TYPE2); handle.synchronize()
    check_against(po, out, ref, oracle.coefficients(ref["R"], ref["Q"]), None, tol_b=1e-9, tol_col=1e-8, tol_inv=1e-7)


def test_gram_fp32_tall(handle, oracle, po):
    """GRAM plan on fp32 storage at a size where the SYRK grid is fully populated, ragged row count"""
    import pls_amd
    torch = _torch()
    N, K, M, A = 70001 * 4 // 4 + 3, 300, 2, 6          # N % 4 != 0: the last slab is ragged
    X = handle.synth_x(0, N, K, 31, dtype=torch.float32); Y = handle.synth_y(0, N, M, 31, dtype=torch.float32)
    handle.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_KERNEL)
    ref = handle.fit_device(X, Y, A); handle.synchronize()
    handle.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_GRAM)
    out = handle.fit_device(X, Y, A); handle.synchronize()
    handle.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_KERNEL)
    assert po.rel_fro(out["B"].cpu().numpy(), ref["B"].cpu().numpy()) < 2e-5
    Xh = X[:2000].cpu().numpy().astype(np.float64)
    assert po.rel_fro(out["T"][:2000, 0].cpu().numpy().astype(np.float64), Xh @ out["R"][:, 0].cpu().numpy()) < 1e-5


def test_fp32_storage(handle, oracle, po, mode):
    """BASELINE config 4 is fp32 (no reference counterpart: float_type is double, pls.h:22).
    fp32 storage of X, Y, T with fp64 accumulation, checked against the fp64 oracle run on the
    same fp32-rounded inputs.  Tolerance 2e-5 on B (T is rounded to fp32 between the passes)."""
    torch = _torch()
    N, K, M, A = 4096, 200, 8, 10
    X = handle.synth_x(0, N, K, 7, dtype=torch.float32); Y = handle.synth_y(0, N, M, 7, dtype=torch.float32)
    Xh = X.cpu().numpy().astype(np.float64); Yh = Y.cpu().numpy().astype(np.float64)
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    out = handle.fit_device(X, Y, A); handle.synchronize()
    check_against(po, out, ref, Bref, ref["T"], tol_b=2e-5, tol_col=2e-5, col_err=cerr, tol_inv=1e-4)


# ------------------------------------------------------------------------------------------
# callers either side of the path: pre-processing (f2) and per-component metrics (f3)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,K", [(10, 15), (60, 401), (4097, 33), (100000, 64)])
def test_device_z_scores(handle, oracle, po, N, K):
    torch = _torch()
    rng = np.random.default_rng(5)
    Xh = np.asfortranarray(oracle.synth_x(0, N, K) * rng.uniform(0.1, 30, K) + rng.uniform(-50, 50, K))
    Z, mean, sd = handle.colwise_z_scores(to_dev(Xh)); handle.synchronize()
    Zr = oracle.z_scores(Xh)
    assert np.allclose(mean.cpu().numpy(), Xh.mean(0), rtol=1e-12, atol=1e-12)
    assert np.allclose(sd.cpu().numpy(), Xh.std(0, ddof=1), rtol=1e-11)
    assert np.abs(Z.cpu().numpy() - Zr).max() < 1e-9
    # in place, and the reference's constant-column behaviour (NaN, src/pls.cpp:103)
    Xc = Xh.copy(); Xc[:, 2] = 3.25
    Xd = to_dev(Xc)
    Z2, _, sd2 = handle.colwise_z_scores(Xd, inplace=True); handle.synchronize()
    assert Z2.data_ptr() == Xd.data_ptr() and float(sd2[2]) == 0.0
    z = Z2.cpu().numpy()
    assert np.isnan(z[:, 2]).all() and np.isfinite(np.delete(z, 2, axis=1)).all()
    # fp32 storage
    Z32, m32, _ = handle.colwise_z_scores(to_dev(Xh.astype(np.float32))); handle.synchronize()
    assert np.abs(Z32.cpu().numpy() - oracle.z_scores(Xh.astype(np.float32).astype(np.float64))).max() < 5e-5


@pytest.mark.parametrize("N,K", [(70001, 40), (513, 7), (3, 5), (1, 4), (200003, 33)])
def test_one_sweep_statistics_are_stable(handle, N, K):
    """mean and sd from ONE sweep (colmoments_kernel) on columns that break a naive sum-of-squares: offsets of 1e8 sd, an
    outlier in the first row, a constant column; against the two-pass statistics in extended precision (src/pls.cpp:69-83
    is two-pass) and against the two-pass kernels of the sharded form"""
    rng = np.random.default_rng(N + K)
    Xh = rng.standard_normal((N, K))
    Xh[:, 0] += 1e8
    Xh[:, 1 % K] = Xh[:, 1 % K] * 1e-3 - 4e6
    Xh[0, 2 % K] = 1e7                        # the first row an outlier of 1e7 sd
    if K > 3:
        Xh[:, 3] = -17.5
    Xh = np.asfortranarray(Xh)
    Z, mean, sd = handle.colwise_z_scores(to_dev(Xh)); handle.synchronize()
    xl = Xh.astype(np.longdouble)
    mr = xl.mean(0)
    sr = np.sqrt(((xl - mr) ** 2).sum(0) / (N - 1)) if N > 1 else np.full(K, np.nan)
    assert np.allclose(mean.cpu().numpy(), mr.astype(np.float64), rtol=1e-13, atol=1e-15)
    got = sd.cpu().numpy()
    if N > 1:
        assert np.allclose(got, sr.astype(np.float64), rtol=1e-12, atol=0), np.nanmax(np.abs(got / sr.astype(np.float64) - 1))
        if K > 3:
            assert got[3] == 0.0
        zr = ((xl - mr) / sr).astype(np.float64)
        ok = np.isfinite(zr)
        assert np.abs(Z.cpu().numpy()[ok] - zr[ok]).max() < 1e-7   # (x - mean) itself is rounded at 1e8 * 2^-53
        # the scale pass forms the quotient from a reciprocal and one residual step: correctly rounded, i.e. the bits of
        # the reference's (x - mean) / sd (src/pls.cpp:103) for the same mean and sd
        with np.errstate(divide="ignore", invalid="ignore"):
            zq = (Xh - mean.cpu().numpy()) / got
        assert np.array_equal(Z.cpu().numpy(), zq, equal_nan=True)
    else:
        assert np.isnan(got).all()


def _random_shapes(count, seed):
    rng = np.random.default_rng(seed)
    shapes = []
    for _ in range(count):
        K = int(rng.choice([1, 2, 3, 5, 17, 31, 32, 33, 63, 64, 65, 100, 255, 257, 512, 700, 1025, 1300]))
        N = int(rng.choice([1, 2, 7, 31, 32, 33, 63, 64, 65, 127, 129, 500, 1023, 1025, 2047, 4099, 20001]))
        M = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 9, 16, 31]))
        A = int(rng.integers(1, max(2, min(K, N - 1 if N > 1 else 1, 12) + 1)))
        shapes.append((N, K, M, min(A, K)))
    return shapes


@pytest.mark.parametrize("N,K,M,A", _random_shapes(24, 20261003))
def test_random_shapes_all_plans(handle, oracle, po, mode, N, K, M, A):
    """Seeded random shapes around every tile boundary of the kernels (32/64 rows, 32 column groups, 1024-column
    resident tile, M = 1 / 2-8 / 9-32 direction solves), every plan, against the oracle."""
    Xh, Yh = oracle.synth_x(0, N, K, seed=5 + N * 131 + K), oracle.synth_y(0, N, M, seed=5 + N * 131 + K)
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    if not np.isfinite(Bref).all():  # degenerate draw (rank exhausted): NaN behaviour is covered elsewhere
        pytest.skip("rank-deficient draw")
    out = handle.fit_device(to_dev(Xh), to_dev(Yh), A); handle.synchronize()
    # B is held at the north star's 1e-10 whatever the conditioning of the individual components (B depends on
    # the subspace the first A weight vectors span, not on the vectors); the per-component tolerance of W, P, Q, R, T
    # follows the disagreement of the two independent fp64 CPU routes on that component, capped at 1e-3 -- beyond
    # that a component's direction carries no information and only B and the invariants are checked.
    got = out["B"].cpu().numpy()
    assert np.isfinite(got).all()
    assert po.rel_fro(got, Bref) < TOL_B
    if np.max(cerr) > 1e-3:
        return
    check_against(po, out, ref, Bref, Tref=ref["T"], col_err=cerr, tol_b=TOL_B, tol_inv=1e-6)


@pytest.mark.parametrize("N,K,M,A", [(10, 15, 2, 2), (60, 401, 1, 10), (5000, 40, 3, 7), (1 << 18, 64, 2, 20),
                                     (3000, 400, 3, 400)])   # A*M = 1200 running sums: swept in two component ranges
def test_sse_by_components(handle, oracle, po, N, K, M, A):
    import pls_amd
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    X, Y = to_dev(Xh), to_dev(Yh)
    m = pls_amd.Model(X, Y, pls_amd.KERNEL_TYPE1, A, handle=handle)
    ev, sse = m.explained_variance_by_components(X, Y); handle.synchronize()
    ref = oracle.plsr(Xh, Yh, A)
    for c in (range(1, A + 1) if A <= 20 else (1, 2, 100, 341, 342, 343, A)):
        evr, sser = po.explained_variance(Xh, Yh, ref["R"], ref["Q"], c)
        assert np.allclose(sse[:, c - 1].cpu().numpy(), sser, rtol=1e-8 if A <= 20 else 1e-6)
        assert np.allclose(ev[:, c - 1].cpu().numpy(), evr, rtol=1e-7 if A <= 20 else 1e-5, atol=1e-10)
        # and it agrees with the reference's route (one X*B_c pass per component count)
        assert np.allclose(m.SSE(X, Y, c).cpu().numpy(), sser, rtol=1e-8 if A <= 20 else 1e-6)


# ------------------------------------------------------------------------------------------
# error behaviour at the boundary (reference: asserts only, src/pls.cpp:345-347)
# ------------------------------------------------------------------------------------------
def test_bad_arguments(handle):
    import pls_amd
    torch = _torch()
    X = handle.synth_x(0, 50, 5, 1); Y = handle.synth_y(0, 50, 1, 1)
    with pytest.raises(pls_amd.PlsHipError) as e:
        handle.fit_device(X, Y, 6)  # A > K
    assert e.value.code == 1
    with pytest.raises(pls_amd.PlsHipError) as e:
        handle.fit_device(X, handle.synth_y(0, 50, 1025, 1), 2)  # more than 1024 responses
    assert e.value.code == 4
    # the handle stays usable after an error
    out = handle.fit_device(X, Y, 2); handle.synchronize()
    assert torch.isfinite(out["B"]).all()


@pytest.mark.parametrize("N,K,M,A", [(1, 3, 1, 1), (40, 1, 1, 1), (2, 2, 2, 2), (300, 20, 32, 3), (64, 600, 2, 5)])
def test_extreme_shapes(handle, oracle, po, mode, N, K, M, A):
    """single row, single predictor, the widest supported response block (m = 32), K > N."""
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    out = handle.fit_device(to_dev(Xh), to_dev(Yh), A); handle.synchronize()
    check_against(po, out, ref, Bref, ref["T"], col_err=cerr, tol_inv=1e-7)


@pytest.mark.parametrize("N,K,M,A", [(300, 40, 33, 4), (500, 64, 64, 5), (200, 30, 100, 3), (1000, 513, 40, 6)])
def test_many_responses(handle, oracle, po, mode, N, K, M, A):
    """M > 32 responses: the reference solves any M x M eigenproblem (src/pls.cpp:405-408); here the M-sized work of
    the component update moves to global memory (largem_kernels.hpp) -- slower, same results.  M > K included."""
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    # the generator repeats its 2^-(j mod 16) column scaling every 16 responses: break the ties between the copies
    Yh = np.asfortranarray(Yh * (1.0 + 0.37 * (np.arange(M) // 16))[None, :])
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    out = handle.fit_device(to_dev(Xh), to_dev(Yh), A); handle.synchronize()
    check_against(po, out, ref, Bref, ref["T"], col_err=cerr, tol_inv=1e-7)


@pytest.mark.parametrize("N,K,C", [(64, 4, 9), (1000, 33, 16), (4099, 130, 17), (777, 63, 32), (2048, 257, 33), (300, 1025, 50),
                                   (5001, 100, 64), (1030, 37, 49), (2049, 515, 200), (130, 70, 65)])
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_xb_many_columns(handle, N, K, C, dt):
    """X * B with more than 8 columns: fp32 storage -- and fp64 beyond 32 columns -- runs on the matrix cores (xb_mfma_lds_kernel: 16-row MFMA tiles x
    V row sets, ragged last rows, K not a multiple of the 4-column step, column blocks of 16), fp64 storage on the
    LDS-staged kernel.  Against torch in fp64."""
    torch = _torch()
    dtype = torch.float32 if dt == "f32" else torch.float64
    X = handle.synth_x(0, N, K, 21, dtype=dtype)
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    Bm = torch.randn(K, C, generator=g, dtype=torch.float64).cuda()
    got = handle.xb(X, Bm); handle.synchronize()
    ref = X.double() @ Bm
    err = float((got.double() - ref).norm() / ref.norm())
    assert err < (2e-7 if dt == "f32" else 1e-14), err


@pytest.mark.parametrize("N,K,C,dt", [(262144 + 37, 70, 20, "f64"), (262144, 64, 5, "f64"), (300000, 33, 19, "f64"), (270001, 130, 32, "f64"),
                                      (262144 + 31, 32, 8, "f64"), (524288 + 5, 33, 24, "f32"), (524288, 40, 5, "f32"), (600001, 18, 13, "f32"),
                                      # the windowed form (xb_mfma4w.hpp): Bm beyond LDS, few row tiles (tile slots x sub-windows), partial last tile
                                      (131072, 4096, 8, "f32"), (131072 + 3, 1030, 20, "f64"), (262144, 1024, 32, "f64"), (40001, 200, 19, "f64"),
                                      (3001, 3000, 7, "f64"), (70000, 130, 24, "f32"), (16388, 2052, 12, "f32"), (65536 + 33, 640, 5, "f64"),
                                      (20000, 2000, 20, "f32"), (8200, 1500, 9, "f64"), (16390, 700, 17, "f32"),
                                      # very short: the columns split over workgroups as well, partial sums + finish kernel
                                      (2000, 20000, 20, "f64"), (512, 50000, 8, "f32"), (1001, 9001, 5, "f64"), (130, 70000, 23, "f32"), (64, 140000, 32, "f64")])
def test_xb_tall_5_to_32_columns(handle, N, K, C, dt):
    """X * B with 5..32 columns on a TALL matrix (scores T = X R with A columns, src/pls.cpp:439-442): the 4 x 4 x 4 MFMA kernel
    (xb_mfma4.hpp) -- K not a multiple of the 4-column step or of the batch, column counts that are not multiples of 4, the
    rows beyond the last full tile (one wave's guarded walk), odd N.  Against torch in fp64 on a sample of rows that includes
    the last ones, and against the older kernels (PLS_HIP_XB4=0) on everything."""
    torch = _torch()
    dtype = torch.float32 if dt == "f32" else torch.float64
    X = handle.synth_x(0, N, K, 23, dtype=dtype)
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    Bm = torch.randn(K, C, generator=g, dtype=torch.float64).cuda()
    got = handle.xb(X, Bm); handle.synchronize()
    idx = torch.cat([torch.randint(0, N, (8192,), device="cuda"), torch.arange(max(0, N - 200), N, device="cuda"), torch.arange(0, min(200, N), device="cuda")])
    if N * K > 1 << 27: idx = idx[::8]
    ref = X[idx].double() @ Bm
    tol = 2e-7 if dt == "f32" else 1e-14 * max(1.0, (K / 2000.0) ** 0.5)  # (two summation orders over K terms)
    assert float((got[idx].double() - ref).norm() / ref.norm()) < tol
    with handle_with_env(PLS_HIP_XB4=0) as h0:
        old = h0.xb(X, Bm); h0.synchronize()
        assert float((got.double() - old.double()).norm() / old.double().norm()) < tol
        assert bool(torch.isfinite(got).all())


@pytest.mark.parametrize("N,K,M,A", [(600, 1100, 2, 5), (257, 1500, 1, 4), (1030, 2048, 1, 6), (130, 2500, 2, 5),
                                     (96, 4096, 1, 4), (66, 4100, 2, 4), (130, 2500, 2, 1), (130, 2500, 2, 2),
                                     (130, 2500, 2, 3), (1030, 700, 1, 2), (1030, 700, 1, 3), (2050, 1024, 3, 4)])
def test_wide_matrix(handle, oracle, po, mode, N, K, M, A):
    """K beyond the 256-byte-segment resident tile (K > 1024).  NIPALS plan: short-tile fused pass on the working
    copy for K <= 2048 (16-row fp64 tiles) and K <= 4096 (8-row tiles) from component 2 on, the semi-fused
    deflate+score sweep before that, for odd N (257: no 16-byte row packs) and beyond 4096 columns; KERNEL plan: the
    one-product kernels.  Same results."""
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    out = handle.fit_device(to_dev(Xh), to_dev(Yh), A); handle.synchronize()
    check_against(po, out, ref, Bref, ref["T"], col_err=cerr)


@pytest.mark.parametrize("N,K,M,A", [(300, 2048, 8, 6), (260, 8192, 2, 5), (200, 4100, 4, 7), (130, 16384, 2, 4),
                                     (400, 4096, 8, 70), (300, 5000, 5, 34), (96, 16384, 8, 3)])
def test_cooperative_component_update(handle, oracle, po, mode, N, K, M, A):
    """2 <= M <= 8 with K*M >= 16K values: the component update runs on K/256 workgroups with two in-launch exchanges
    (coop_update.hpp) -- ragged last workgroup (K = 4100, 5000), the 64-workgroup maximum (K = 16384), more components
    than one 32-value chunk of c_j = p_j^T w holds (A = 34, 70), every plan, against the oracle."""
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    out = handle.fit_device(to_dev(Xh), to_dev(Yh), A); handle.synchronize()
    check_against(po, out, ref, Bref, ref["T"], col_err=cerr, tol_inv=1e-7)


@pytest.mark.parametrize("N,K,M,A", [(300, 2048, 16, 5), (260, 1024, 32, 4), (500, 1900, 9, 6), (130, 5000, 12, 4)])
def test_nine_to_32_responses_on_many_columns(handle, oracle, po, mode, N, K, M, A):
    """9 <= M <= 32 with K M >= 16 K values: the component update runs as the multi-workgroup kernels of the
    many-response path (X^T Y deflation, Gram matrix by the column-reduction kernels, r recurrence) with the direction
    solved in one workgroup's LDS -- the one-workgroup kernel took 208 us per component at K = 4096, M = 16."""
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    out = handle.fit_device(to_dev(Xh), to_dev(Yh), A); handle.synchronize()
    check_against(po, out, ref, Bref, ref["T"], col_err=cerr)


@pytest.mark.parametrize("N,K,M,A", [(512, 2300, 3, 4), (516, 1536, 1, 5), (260, 4096, 2, 5)])
def test_wide_matrix_fp32(handle, oracle, po, mode, N, K, M, A):
    torch = _torch()
    X = handle.synth_x(0, N, K, 11, dtype=torch.float32); Y = handle.synth_y(0, N, M, 11, dtype=torch.float32)
    Xh = X.cpu().numpy().astype(np.float64); Yh = Y.cpu().numpy().astype(np.float64)
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    out = handle.fit_device(X, Y, A); handle.synchronize()
    check_against(po, out, ref, Bref, ref["T"], tol_b=2e-5, tol_col=2e-5, col_err=cerr, tol_inv=1e-4)


def test_padded_leading_dimensions(handle, oracle, po, mode):
    """ld > rows for X, Y and T (the C-ABI takes explicit leading dimensions)."""
    import ctypes
    import pls_amd
    from pls_amd import _lib as L
    torch = _torch()
    N, K, M, A = 1000, 48, 2, 6
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    X = pls_amd.colmajor_empty(N, K, torch.float64, "cuda", ld=N + 24); X.copy_(torch.from_numpy(Xh))
    Y = pls_amd.colmajor_empty(N, M, torch.float64, "cuda", ld=N + 8); Y.copy_(torch.from_numpy(Yh))
    out = {k: pls_amd.colmajor_empty(K, A, torch.float64, "cuda", ld=K) for k in "WPR"}
    out["Q"] = pls_amd.colmajor_empty(M, A, torch.float64, "cuda", ld=M)
    out["B"] = pls_amd.colmajor_empty(K, M, torch.float64, "cuda", ld=K)
    out["T"] = pls_amd.colmajor_empty(N, A, torch.float64, "cuda", ld=N + 40)
    out["T"].untyped_storage()  # keep a reference
    guard = torch.full((A, N + 40), 7.0, dtype=torch.float64, device="cuda")
    out["T"] = guard[:, :N].t()
    handle.fit_device(X, Y, A, out=out); handle.synchronize()
    check_against(po, out, ref, Bref, ref["T"], col_err=cerr)
    assert bool((guard[:, N:] == 7.0).all()), "the padding between T columns must not be written"


def test_bad_arguments_every_entry_point(handle):
    """each C-ABI entry validates shapes / pointers / enums and reports PLS_HIP_ERR_INVALID (1) or
    UNSUPPORTED (4) without touching the device state"""
    import ctypes
    import pls_amd
    from pls_amd import _lib as L
    torch = _torch()
    lib = L.lib()
    h = handle.h
    X = handle.synth_x(0, 64, 6, 1); Y = handle.synth_y(0, 64, 2, 1)
    out = handle.fit_device(X, Y, 3); handle.synchronize()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    nul = ctypes.c_void_p(0)
    W, P, Q, R, T, B = (out[k] for k in "WPQRTB")
    fit = lambda **kw: lib.pls_hip_fit(h, kw.get("X", p(X)), kw.get("ldx", 64), p(Y), 64, kw.get("N", 64), kw.get("K", 6),
                                        kw.get("M", 2), kw.get("A", 3), kw.get("method", 0), kw.get("dtype", 0),
                                        kw.get("mem", 1), p(W), p(P), p(Q), p(R), p(T), 64, p(B))
    assert fit() == 0
    assert fit(X=nul) == 1 and fit(ldx=10) == 1 and fit(N=0) == 1 and fit(K=0) == 1 and fit(A=0) == 1 and fit(A=7) == 1
    assert fit(method=5) == 1 and fit(dtype=9) == 1 and fit(mem=3) == 1 and fit(M=1025) == 4
    assert b"" != lib.pls_hip_last_error(h)
    assert lib.pls_hip_xb(h, p(X), 64, 64, 6, nul, 6, 2, 0, 1, p(T), 64) == 1
    assert lib.pls_hip_xb(h, p(X), 10, 64, 6, p(R), 6, 2, 0, 1, p(T), 64) == 1
    assert lib.pls_hip_xty(h, p(X), 64, p(Y), 64, 0, 6, 2, 0, p(B)) == 1
    assert lib.pls_hip_deflate(h, p(X), 64, p(X), 64, 64, 6, nul, p(P), 0) == 1
    assert lib.pls_hip_coefficients(h, p(R), p(Q), 6, 2, 3, 4, 1, p(B)) == 1          # comp > A (src/pls.cpp:445)
    assert lib.pls_hip_colwise_z_scores(h, p(X), 64, 64, 10, 6, 0, nul, 64, p(W), p(P)) == 1   # n_total < N
    assert lib.pls_hip_sse_by_components(h, p(T), 64, p(Y), 64, 64, 3, 2, nul, 0, p(B)) == 1
    assert lib.pls_hip_model_sse(h, p(X), 64, p(Y), 64, 64, 6, 2, 3, p(R), p(Q), 0, 7, p(B)) == 1
    idx = (ctypes.c_int64 * 2)(0, 99)
    assert lib.pls_hip_cv_folds(h, p(X), 64, p(Y), 64, 64, 6, 2, 3, idx, 1, 2, 0, 1, p(B)) == 1   # index out of range
    assert lib.pls_hip_cv_folds(h, p(X), 64, p(Y), 64, 64, 6, 2, 3, idx, 1, 2, 5, 1, p(B)) == 1   # bad dtype
    assert lib.pls_hip_set_option(h, 42, 1) == 1 and lib.pls_hip_set_option(h, L.OPT_ALGO, 9) == 1
    assert lib.pls_hip_set_reducer(h, L.ALLREDUCE_FN(0), None, 1, 1) == 1            # rank >= nranks
    assert lib.pls_hip_synth_x(h, p(X), 3, 0, 64, 6, 1, 0) == 1                        # ld < rows
    # the handle is still healthy
    out2 = handle.fit_device(X, Y, 3); handle.synchronize()
    assert torch.equal(out2["B"], out["B"])


def test_rank_deficient_leading_components(handle, oracle, po):
    """A > rank(X): the surplus columns are inf/NaN/garbage in the reference too (division by
    tt ~ 0, src/pls.cpp:427-428); the leading rank(X) components are unaffected."""
    N, K, A = 5, 9, 7
    Xh = oracle.synth_x(0, N, K); Yh = oracle.synth_y(0, N, 1)
    ref = oracle.plsr(Xh, Yh, A)
    out = handle.fit_device(to_dev(Xh), to_dev(Yh), A); handle.synchronize()
    W = out["W"].cpu().numpy()
    s = po.sign_align(ref["W"][:, :4], W[:, :4])
    assert po.rel_fro(W[:, :4] * s, ref["W"][:, :4]) < 1e-8


# ------------------------------------------------------------------------------------------
# BASELINE full size (config 3: 1,048,576 x 512, m = 1) through size-independent properties
# ------------------------------------------------------------------------------------------
def test_full_size_properties(handle, po):
    import pls_amd
    torch = _torch()
    N, K, M, A = 1 << 20, 512, 1, 4
    X = handle.synth_x(0, N, K, pls_amd.SEED_DEFAULT); Y = handle.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
    outs = {}
    for algo, fuses in ((0, (0, 1)), (1, (0, 1)), (2, (1,))):
        for fuse in fuses:
            handle.set_option(pls_amd.OPT_ALGO, algo); handle.set_option(pls_amd.OPT_FUSE, fuse)
            outs[(algo, fuse)] = handle.fit_device(X, Y, A)
            handle.synchronize()
    handle.set_option(pls_amd.OPT_ALGO, 0); handle.set_option(pls_amd.OPT_FUSE, 1)
    base = outs[(0, 0)]
    Bb = base["B"].cpu().numpy()
    for key, o in outs.items():  # the four execution plans agree
        assert po.rel_fro(o["B"].cpu().numpy(), Bb) < TOL_B, key
    W = base["W"].cpu().numpy(); P = base["P"].cpu().numpy(); R = base["R"].cpu().numpy()
    assert np.allclose((W * W).sum(0), 1.0, atol=1e-12)
    assert np.allclose(P.T @ R, np.eye(A), atol=1e-9)
    T = base["T"]
    G = (T.t() @ T).cpu().numpy()
    assert np.abs(G - np.diag(np.diag(G))).max() < 1e-9 * np.diag(G).max()  # orthogonal scores
    # deflation: X - t p^T is orthogonal to t (X^T t = p tt)
    t0 = T[:, 0].contiguous(); p0 = torch.from_numpy(P[:, 0].copy()).cuda()
    D = handle.deflate(X, t0, p0)
    resid = handle.xty(D, t0[:, None]); handle.synchronize()
    assert float(resid.abs().max()) < 1e-9 * float(G[0, 0])
    # row block of the full-size product against the oracle on the same rows
    rows = slice(123456, 123456 + 2048)
    Xs = X[rows].cpu().numpy()
    assert po.rel_fro(T[rows, 0].cpu().numpy(), Xs @ R[:, 0]) < 1e-12


def test_full_size_config4_fp32(handle, po):
    """BASELINE config 4 at full size (131,072 x 4,096 fp32, m = 8): the plans agree with each other,
    scores are orthogonal, P^T R = I -- fp32 storage tolerances."""
    import pls_amd
    torch = _torch()
    N, K, M, A = 131072, 4096, 8, 5
    X = handle.synth_x(0, N, K, pls_amd.SEED_DEFAULT, dtype=torch.float32)
    Y = handle.synth_y(0, N, M, pls_amd.SEED_DEFAULT, dtype=torch.float32)
    outs = {}
    for algo, fuse in ((0, 1), (1, 1), (1, 0)):
        handle.set_option(pls_amd.OPT_ALGO, algo); handle.set_option(pls_amd.OPT_FUSE, fuse)
        outs[(algo, fuse)] = handle.fit_device(X, Y, A)
        handle.synchronize()
    handle.set_option(pls_amd.OPT_ALGO, 0); handle.set_option(pls_amd.OPT_FUSE, 1)
    Bb = outs[(0, 1)]["B"].cpu().numpy()
    assert np.isfinite(Bb).all()
    for key, o in outs.items():
        assert po.rel_fro(o["B"].cpu().numpy(), Bb) < 2e-5, key
    base = outs[(0, 1)]
    P = base["P"].cpu().numpy(); R = base["R"].cpu().numpy()
    assert np.allclose(P.T @ R, np.eye(A), atol=1e-4)
    T = base["T"].to(torch.float64)
    G = (T.t() @ T).cpu().numpy()
    assert np.abs(G - np.diag(np.diag(G))).max() < 1e-4 * np.diag(G).max()
    rows = slice(77777, 77777 + 512)
    assert po.rel_fro(T[rows, 0].cpu().numpy(), X[rows].cpu().numpy().astype(np.float64) @ R[:, 0]) < 1e-5


@pytest.mark.parametrize("N,K,M,A,ts,nf", [(10, 15, 2, 2, 1, 10), (60, 401, 1, 10, 1, 60), (60, 401, 1, 6, 18, 25),
                                           (200, 24, 3, 5, 60, 12)])
def test_batched_cv_folds(handle, oracle, po, N, K, M, A, ts, nf):
    """f4: all cross-validation folds in one launch (Gram downdates) against one oracle refit per fold."""
    if K == 401:
        Xh = oracle.z_scores(po.read_csv(os.path.join(DATA, "nir.csv")))
        Yh = oracle.z_scores(po.read_csv(os.path.join(DATA, "octane.csv")))
    elif K == 15:
        Xh = oracle.z_scores(po.read_csv(os.path.join(DATA, "toyX.csv")))
        Yh = oracle.z_scores(po.read_csv(os.path.join(DATA, "toyY.csv")))
    else:
        Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    rng = np.random.default_rng(7)
    idx = np.arange(N)[:, None] if ts == 1 and nf == N else np.stack([rng.permutation(N)[:ts] for _ in range(nf)])
    E = handle.cv_folds(to_dev(Xh), to_dev(Yh), A, idx).cpu().numpy()       # device path
    Eh = handle.cv_folds(Xh, Yh, A, idx)                                      # host-memory path
    ref = np.zeros_like(Eh)
    for f in range(idx.shape[0]):
        test = idx[f]
        train = np.setdiff1d(np.arange(N), test)
        c = oracle.plsr(Xh[train], Yh[train], A)
        for nc in range(1, A + 1):
            B = oracle.coefficients(c["R"], c["Q"], nc)
            ref[:, f * ts:(f + 1) * ts, nc - 1] = (Yh[test] - Xh[test] @ B).T
    scale = np.abs(ref).max()
    assert np.abs(E - ref).max() < 1e-8 * max(scale, 1.0)
    assert np.abs(Eh - ref).max() < 1e-8 * max(scale, 1.0)


@pytest.mark.parametrize("N,K,M,A,dt", [(700001, 24, 2, 6, "f64"), (90000, 200, 1, 8, "f64"), (90000, 200, 3, 5, "f32"), (50, 12, 1, 3, "f64")])
def test_host_entry_accumulates_gram_during_upload(handle, oracle, po, N, K, M, A, dt):
    """pls_hip_fit on HOST pointers under ALGO_AUTO (and for KERNEL_TYPE2): X^T X and X^T Y are accumulated on the matrix
    cores row block by row block while X crosses PCIe, the component loop never passes over X (no fused / score
    launches), T = X R follows in one pass.  Several 32 MB blocks, a ragged last block, fp32 storage (no 16-byte columns
    when N % 4 != 0: plain upload, the fit forms the products itself)."""
    import pls_amd
    dtype = np.float64 if dt == "f64" else np.float32
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    if dt == "f32":
        Xh = np.asfortranarray(Xh.astype(np.float32).astype(np.float64)); Yh = np.asfortranarray(Yh.astype(np.float32).astype(np.float64))
    ref = oracle.plsr(Xh, Yh, A)
    Bref = oracle.coefficients(ref["R"], ref["Q"])
    tol = 1e-10 if dt == "f64" else 2e-5
    handle.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_AUTO)
    handle.set_option(pls_amd.OPT_PROFILE, 1)
    try:
        handle.timing()
        out = handle.fit_host(Xh, Yh, A, dtype=dtype)
        tm = handle.timing()
        assert po.rel_fro(out["B"], Bref) < tol
        assert po.rel_fro(out["T"].astype(np.float64), Xh @ out["R"]) < (1e-11 if dt == "f64" else 1e-6)
        if N >= 4096:
            assert tm["launches"]["fused"] == 0          # the Gram plan: no pass over X per component
        o2 = handle.fit_host(Xh, Yh, A, method=pls_amd.KERNEL_TYPE2, dtype=dtype)
        assert o2["T"] is None and po.rel_fro(o2["B"], Bref) < tol
        handle.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_KERNEL)
        o3 = handle.fit_host(Xh, Yh, A, dtype=dtype)
        assert po.rel_fro(o3["B"], out["B"]) < tol
    finally:
        handle.set_option(pls_amd.OPT_ALGO, 0)
        handle.set_option(pls_amd.OPT_PROFILE, 0)


def test_roctx_ranges_switch():
    """PLS_HIP_ROCTX=1: every fit is wrapped in roctx ranges (marker library resolved at run time); the fit itself is
    unchanged.  (profiles/r2 holds a rocprofv3 --marker-trace of it.)"""
    import subprocess, sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r); import torch, pls_amd; h = pls_amd.Handle(); "
            "X = h.synth_x(0, 4096, 64, 1); Y = h.synth_y(0, 4096, 2, 1); "
            "a = h.fit_device(X, Y, 5); h.set_option(pls_amd.OPT_ALGO, 2); b = h.fit_device(X, Y, 5); h.synchronize(); "
            "print(float((a['B'] - b['B']).norm() / a['B'].norm()) < 1e-10)") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PLS_HIP_ROCTX="1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("True"), r.stderr[-2000:]


def test_kernel_type2_has_no_scores(handle):
    """T exists for KERNEL_TYPE1 only (reference src/pls.cpp:394,434): the Python mirror returns None, never an
    uninitialised buffer"""
    import pls_amd
    X = handle.synth_x(0, 500, 12, 3); Y = handle.synth_y(0, 500, 1, 3)
    assert handle.fit_device(X, Y, 3, method=pls_amd.KERNEL_TYPE2)["T"] is None
    assert handle.fit_host(X.cpu().numpy(), Y.cpu().numpy(), 3, method=pls_amd.KERNEL_TYPE2)["T"] is None
    m = pls_amd.Model(X, Y, pls_amd.KERNEL_TYPE2, 3, handle=handle)
    assert m.T is None and m.coefficients().shape == (12, 1)


def test_auto_plan_choice(handle):
    """AUTO = GRAM where the cost model says so (tall fp64, many components), KERNEL otherwise; checked
    through which kernel families ran."""
    import pls_amd
    handle.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_AUTO)
    handle.set_option(pls_amd.OPT_PROFILE, 1)
    try:
        X = handle.synth_x(0, 1 << 18, 256, 5); Y = handle.synth_y(0, 1 << 18, 1, 5)
        handle.timing()
        handle.fit_device(X, Y, 20); tm = handle.timing()
        assert tm["launches"]["fused"] == 0 and tm["launches"]["xty"] >= 1     # the SYRK (X^T Y on board), no per-component pass
        handle.fit_device(X, Y, 2); tm = handle.timing()
        assert tm["launches"]["fused"] == 2                                      # two components: passes are cheaper
        Xs = handle.synth_x(0, 500, 64, 5); Ys = handle.synth_y(0, 500, 1, 5)
        handle.fit_device(Xs, Ys, 20); tm = handle.timing()
        assert tm["launches"]["fused"] == 0 and tm["launches"]["xty"] == 0          # small N: the one-launch resident fit
        Xw = handle.synth_x(0, 2000, 600, 5); Yw = handle.synth_y(0, 2000, 1, 5)
        handle.fit_device(Xw, Yw, 20); tm = handle.timing()
        assert tm["launches"]["fused"] == 20                                     # beyond its 416 columns: KERNEL
    finally:
        handle.set_option(pls_amd.OPT_ALGO, 0)
        handle.set_option(pls_amd.OPT_PROFILE, 0)


def test_very_tall_64bit_indexing(handle, po):
    """N = 2^24 rows x 64 columns (8.6 GB): element and byte offsets beyond 2^31 / 2^32 on every path; the
    resident-tile kernels decline this shape (column-group span > 2 GiB) and the one-product kernels
    take over.  Size-independent checks: the plans agree, scores orthogonal, a row block of T against
    the host product."""
    import pls_amd
    torch = _torch()
    N, K, M, A = 1 << 24, 64, 2, 4
    X = handle.synth_x(0, N, K, 3); Y = handle.synth_y(0, N, M, 3)
    outs = {}
    for algo in (0, 1, 2):
        handle.set_option(pls_amd.OPT_ALGO, algo)
        outs[algo] = handle.fit_device(X, Y, A)
        handle.synchronize()
    handle.set_option(pls_amd.OPT_ALGO, 0)
    Bb = outs[0]["B"].cpu().numpy()
    assert np.isfinite(Bb).all()
    for algo, o in outs.items():
        assert po.rel_fro(o["B"].cpu().numpy(), Bb) < TOL_B, algo
    T = outs[0]["T"]
    G = (T.t() @ T).cpu().numpy()
    assert np.abs(G - np.diag(np.diag(G))).max() < 1e-9 * np.diag(G).max()
    R = outs[0]["R"].cpu().numpy()
    for r0 in (0, (1 << 23) + 12345, N - 1000):       # first rows, beyond 2^32 bytes into a column set, last rows
        rows = slice(r0, r0 + 1000)
        assert po.rel_fro(T[rows, 1].cpu().numpy(), X[rows].cpu().numpy() @ R[:, 1]) < 1e-11
    # the last column of X sits beyond 8 GB: its generator values must match the host twin
    assert np.array_equal(X[N - 4:, K - 1].cpu().numpy(), po.synth_x(N - 4, 4, K, 3)[:, K - 1])


def test_model_api_cross_validation(handle, oracle, po):
    """the Python mirror's cv_LOO / cv_LSO / cv_NEW_DATA (reference pls.h:235-241) on the nir example"""
    import pls_amd
    Xh = oracle.z_scores(po.read_csv(os.path.join(DATA, "nir.csv")))
    Yh = oracle.z_scores(po.read_csv(os.path.join(DATA, "octane.csv")))
    A = 5
    m = pls_amd.Model(to_dev(Xh), to_dev(Yh), pls_amd.KERNEL_TYPE1, A, handle=handle)
    E = m.cv_LOO().cpu().numpy()
    assert E.shape == (1, 60, A)
    i = 17
    keep = np.arange(60) != i
    c = oracle.plsr(Xh[keep], Yh[keep], A)
    for nc in range(1, A + 1):
        assert abs(E[0, i, nc - 1] - (Yh[i] - Xh[i] @ oracle.coefficients(c["R"], c["Q"], nc))[0]) < 1e-9
    rmse = np.sqrt((E ** 2).mean(axis=1))
    assert np.allclose(rmse[0, :3], [0.849836, 0.501343, 0.160829], atol=2e-6)     # the CLI's printed LOO RMSE
    Es = m.cv_LSO(0.3, 7, np.random.default_rng(1)).cpu().numpy()
    assert Es.shape == (1, 7 * 18, A) and np.isfinite(Es).all()
    ref = oracle.plsr(Xh, Yh, A)
    En = m.cv_NEW_DATA(to_dev(Xh[:9]), to_dev(Yh[:9])).cpu().numpy()
    for nc in range(1, A + 1):
        want = Yh[:9] - Xh[:9] @ oracle.coefficients(ref["R"], ref["Q"], nc)
        assert np.abs(En[:, :, nc - 1].T - want).max() < 1e-9
    # host-memory model
    mh = pls_amd.Model(Xh, Yh, pls_amd.KERNEL_TYPE1, A, handle=handle)
    assert np.abs(mh.cv_LOO() - E).max() < 1e-9


def test_many_components_equals_least_squares(handle, oracle, po, mode):
    """A = K = 200 components (the reference's CV inner models fit K components, src/pls.cpp:356-359):
    the full-rank fit reproduces ordinary least squares, and the run exercises the multi-workgroup r update."""
    N, K, M = 3000, 200, 2
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    out = handle.fit_device(to_dev(Xh), to_dev(Yh), K); handle.synchronize()
    B = out["B"].cpu().numpy()
    Bls = np.linalg.lstsq(Xh, Yh, rcond=None)[0]
    assert np.isfinite(B).all() and po.rel_fro(B, Bls) < 1e-7
    # the leading scores are mutually orthogonal (the trailing ones carry ~1e-13 of the variance: rounding noise)
    T = out["T"].cpu().numpy()[:, :20]
    G = T.T @ T
    d = np.sqrt(np.diag(G))
    assert np.abs(G / np.outer(d, d) - np.eye(20)).max() < 1e-8


def test_kernel_type2_beyond_16384_columns(handle, oracle, po):
    """KERNEL_TYPE2 at K = 20,000 (a 3.2 GB X^T X, 25.6 GB of reduction slices): same B as KERNEL_TYPE1 on the same data
    and as the oracle's method 1."""
    import pls_amd
    N, K, M, A = 96, 20000, 1, 4
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    ref = oracle.plsr(Xh, Yh, A)
    Bref = oracle.coefficients(ref["R"], ref["Q"])
    X, Y = to_dev(Xh), to_dev(Yh)
    o2 = handle.fit_device(X, Y, A, method=pls_amd.KERNEL_TYPE2); handle.synchronize()
    assert po.rel_fro(o2["B"].cpu().numpy(), Bref) < TOL_B
    o1 = handle.fit_device(X, Y, A); handle.synchronize()
    assert po.rel_fro(o1["B"].cpu().numpy(), Bref) < TOL_B


def test_more_than_4096_components(handle, po):
    """A > 4096 (hence K > 4096): components are computed strictly in sequence, so the leading ones equal those of a
    short fit bit for bit, everything stays finite, P^T R = I on the leading block, the leading scores are orthogonal."""
    torch = _torch()
    N, K, M, A = 4400, 4200, 1, 4150
    X = handle.synth_x(0, N, K, 17); Y = handle.synth_y(0, N, M, 17)
    short = {k: v.clone() for k, v in handle.fit_device(X, Y, 12).items()}; handle.synchronize()
    out = handle.fit_device(X, Y, A); handle.synchronize()
    for k in "WPQR":
        assert torch.equal(out[k][:, :12], short[k]), k
    assert torch.equal(out["T"][:, :12], short["T"])
    assert torch.isfinite(out["B"]).all() and torch.isfinite(out["R"][:, :200]).all()
    P = out["P"][:, :40].cpu().numpy(); R = out["R"][:, :40].cpu().numpy()
    assert np.allclose(P.T @ R, np.eye(40), atol=1e-7)
    T = out["T"][:, :40].cpu().numpy()
    G = T.T @ T; d = np.sqrt(np.diag(G))
    assert np.abs(G / np.outer(d, d) - np.eye(40)).max() < 1e-8


@pytest.mark.parametrize("N,K,A,dt,pad,M", [(1025, 26, 5, "f64", 0, 1), (5000, 128, 10, "f64", 0, 1), (4001, 416, 6, "f64", 3, 1), (20000, 16, 5, "f64", 1, 1),
                                            (100000, 40, 12, "f64", 0, 1), (262144, 26, 4, "f64", 0, 1), (3001, 77, 7, "f32", 5, 1), (70000, 50, 9, "f32", 0, 1),
                                            (1024, 40, 6, "f64", 0, 1), (1025, 26, 5, "f64", 0, 3), (5000, 128, 8, "f64", 2, 4), (4001, 416, 5, "f64", 0, 2),
                                            (100000, 40, 12, "f64", 0, 8), (3001, 77, 7, "f32", 1, 5), (2000, 300, 4, "f64", 0, 8)])
def test_resident_single_launch_fit(handle, oracle, po, N, K, A, dt, pad, M):
    """Mid-size fits of 1..8 responses run as ONE launch on up to 256 workgroups (resident_kernels.hpp): every workgroup keeps its
    rows of X in registers, one grid-wide exchange per component.  Every rows-per-workgroup shape (64 ... 1024), ragged last
    workgroups, a padded leading dimension, fp32 storage: results against the oracle and the general plan (PLS_HIP_RESIDENT=0),
    one launch in all, repeated fits equal bit for bit."""
    import pls_amd
    torch = _torch()
    tdt = torch.float64 if dt == "f64" else torch.float32
    Xbig = torch.zeros((K, N + pad), dtype=tdt, device="cuda")
    Xbig[:, :N] = handle.synth_x(5, N, K, 31, dtype=tdt).T
    Xd = Xbig.T[:N]                                   # column-major view, ld = N + pad
    Yd = handle.synth_y(5, N, M, 31, dtype=tdt)
    Xh = np.asfortranarray(Xd.cpu().numpy().astype(np.float64)); Yh = np.asfortranarray(Yd.cpu().numpy().astype(np.float64))
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    handle.set_option(pls_amd.OPT_PROFILE, 2)
    try:
        handle.timing()
        out = handle.fit_device(Xd, Yd, A)
        handle.synchronize()
        t = handle.timing()
        assert sum(t["launches"].values()) == 1, t["launches"]
        first = {k: v.clone() for k, v in out.items()}
        for _ in range(5):
            again = handle.fit_device(Xd, Yd, A); handle.synchronize()
            for k in "WPQRTB":
                assert torch.equal(again[k], first[k]), k
        with handle_with_env(PLS_HIP_RESIDENT=0) as general:
            general.set_option(pls_amd.OPT_PROFILE, 2)
            plain = general.fit_device(Xd, Yd, A)
            general.synchronize()
            assert sum(general.timing()["launches"].values()) > 1
            plain = {k: v.clone() for k, v in plain.items()}
    finally:
        handle.set_option(pls_amd.OPT_PROFILE, 0)
    tol = dict(tol_b=1e-10, tol_col=1e-9) if dt == "f64" else dict(tol_b=2e-5, tol_col=2e-4, tol_inv=1e-3)
    check_against(po, first, ref, Bref, Tref=ref["T"], col_err=cerr, **tol)
    assert po.rel_fro(first["B"].cpu().numpy(), plain["B"].cpu().numpy()) < (1e-10 if dt == "f64" else 2e-5)


@pytest.mark.parametrize("N,K,A,dt,pad,M", [(1025, 26, 5, "f64", 0, 1), (5000, 128, 10, "f64", 0, 1), (5003, 128, 11, "f64", 3, 1), (20000, 16, 5, "f64", 1, 1),
                                            (100000, 40, 12, "f64", 0, 1), (262144, 26, 4, "f64", 0, 1), (3001, 77, 7, "f32", 5, 1), (70000, 50, 9, "f32", 0, 1),
                                            (2050, 64, 32, "f64", 0, 1), (1500, 100, 20, "f64", 0, 1), (8000, 100, 5, "f64", 0, 1), (300, 90, 6, "f64", 0, 1),
                                            (1025, 26, 5, "f64", 0, 3), (5000, 100, 8, "f64", 2, 4), (100000, 40, 12, "f64", 0, 8), (3001, 77, 7, "f32", 1, 5),
                                            (2000, 120, 6, "f64", 0, 2), (9000, 128, 9, "f64", 0, 7), (7001, 96, 6, "f32", 3, 3), (5001, 112, 7, "f64", 0, 2),
                                            (4000, 48, 5, "f64", 0, 1), (100000, 64, 8, "f64", 0, 1)])
def test_resident_gram_single_launch_fit(handle, oracle, po, N, K, A, dt, pad, M):
    """AUTO on mid-size data (1..8 responses) with at most 128 columns: ONE launch with three grid-wide hand-offs whatever A
    (resident_gram.hpp: X^T X and X^T Y on the matrix cores, the component loop on XX in one workgroup's LDS, the scores at the
    end).  Row counts that leave ragged last workgroups, a padded leading dimension, fp32 storage, the direct and the sliced
    sum of the parts, both forms of the first phase (48 columns and more with 16-byte-aligned columns: a 16 x 16 block of XX per
    workgroup and row split; an odd leading dimension keeps the row form), as many components as the LDS holds: against the oracle and the general plan, one launch in all, repeated
    fits equal bit for bit; PLS_HIP_RESIDENT_GRAM=0 gives the per-component resident kernel."""
    import pls_amd
    torch = _torch()
    tdt = torch.float64 if dt == "f64" else torch.float32
    Xbig = torch.zeros((K, N + pad), dtype=tdt, device="cuda")
    Xbig[:, :N] = handle.synth_x(5, N, K, 31, dtype=tdt).T
    Xd = Xbig.T[:N]                                   # column-major view, ld = N + pad
    Yd = handle.synth_y(5, N, M, 31, dtype=tdt)
    Xh = np.asfortranarray(Xd.cpu().numpy().astype(np.float64)); Yh = np.asfortranarray(Yd.cpu().numpy().astype(np.float64))
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    handle.set_option(pls_amd.OPT_PROFILE, 2)
    handle.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_AUTO)
    try:
        handle.timing()
        out = handle.fit_device(Xd, Yd, A)
        handle.synchronize()
        t = handle.timing()
        assert sum(t["launches"].values()) == 1, t["launches"]
        first = {k: v.clone() for k, v in out.items()}
        for _ in range(5):
            again = handle.fit_device(Xd, Yd, A); handle.synchronize()
            for k in "WPQRTB":
                assert torch.equal(again[k], first[k]), k
        with handle_with_env(PLS_HIP_RESIDENT=0) as general:
            plain = general.fit_device(Xd, Yd, A)
            general.synchronize()
            plain = {k: v.clone() for k, v in plain.items()}
    finally:
        handle.set_option(pls_amd.OPT_PROFILE, 0)
        handle.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_KERNEL)
    tol = dict(tol_b=1e-10, tol_col=1e-9) if dt == "f64" else dict(tol_b=2e-5, tol_col=2e-4, tol_inv=1e-3)
    if A >= 20: tol = dict(tol_b=1e-8, tol_col=1e-7)  # (the later directions of a long fit: XX squares the condition number)
    check_against(po, first, ref, Bref, Tref=ref["T"], col_err=cerr, **tol)
    assert po.rel_fro(first["B"].cpu().numpy(), plain["B"].cpu().numpy()) < (1e-8 if A >= 20 else 1e-10 if dt == "f64" else 2e-5)


def test_resident_gram_fit_wait_that_runs_out_is_reported():
    """The same bounded waits in the X^T X form (AUTO): a limit of one tick ends the launch with the device error of
    pls_hip_synchronize, and the next fit works."""
    code = '''
import os, sys
sys.path.insert(0, %r)
import torch, pls_amd
h = pls_amd.Handle(); h.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_AUTO)
X = h.synth_x(0, 60000, 40, 3); Y = h.synth_y(0, 60000, 1, 3)
os.environ["PLS_HIP_TEST_RESIDENT_LIMIT_TICKS"] = "1"
out = h.fit_device(X, Y, 6)
try:
    h.synchronize(); print("no error")
except pls_amd.PlsHipError as e:
    print("error", e.code)
del os.environ["PLS_HIP_TEST_RESIDENT_LIMIT_TICKS"]
out = h.fit_device(X, Y, 6); h.synchronize()
print("refit finite", bool(torch.isfinite(out["B"]).all()))
''' % ROOT
    env = dict(os.environ, PLS_AMD_LIBRARY=os.path.join(ROOT, "pls_amd", "csrc", "testing", "libpls_hip.so"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "error 2" in r.stdout and "refit finite True" in r.stdout, r.stdout[-1500:]


def test_resident_fit_wait_that_runs_out_is_reported():
    """The resident fit's workgroups wait for each other inside the launch; a wait beyond its time limit (another process holding
    the GPU's CUs) must end, poison the results and be reported by pls_hip_synchronize -- and the next fit must work.  The limit is
    set to one tick through the test build of the library (PLS_HIP_TEST_RESIDENT_LIMIT_TICKS is read by testing/libpls_hip.so only)."""
    code = '''
import os, sys
sys.path.insert(0, %r)
import torch, pls_amd
h = pls_amd.Handle()
X = h.synth_x(0, 60000, 40, 3); Y = h.synth_y(0, 60000, 1, 3)
os.environ["PLS_HIP_TEST_RESIDENT_LIMIT_TICKS"] = "1"
out = h.fit_device(X, Y, 6)
try:
    h.synchronize(); print("no error")
except pls_amd.PlsHipError as e:
    print("error", e.code)
del os.environ["PLS_HIP_TEST_RESIDENT_LIMIT_TICKS"]
out = h.fit_device(X, Y, 6); h.synchronize()
print("refit finite", bool(torch.isfinite(out["B"]).all()))
''' % ROOT
    env = dict(os.environ, PLS_AMD_LIBRARY=os.path.join(ROOT, "pls_amd", "csrc", "testing", "libpls_hip.so"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "error 2" in r.stdout and "refit finite True" in r.stdout, r.stdout[-1500:]


@pytest.mark.parametrize("N,K,A,dt,pad", [(60, 401, 10, "f64", 0), (64, 416, 12, "f64", 3), (10, 15, 3, "f64", 0), (65, 200, 8, "f64", 1),
                                          (130, 120, 9, "f64", 0), (1024, 26, 6, "f64", 0), (1000, 20, 6, "f32", 8), (1, 5, 1, "f64", 0)])
def test_single_launch_fit(handle, oracle, po, N, K, A, dt, pad):
    """Small single-response fits run as ONE launch (tiny_kernels.hpp): X in registers, everything K-sized in LDS.
    Every row-block / column-slice shape of that kernel (N <= 64 ... 1024), a padded leading dimension, fp32 storage;
    results against the oracle, and against the three-launches-per-component plan (PLS_HIP_TINY=0)."""
    import pls_amd
    torch = _torch()
    if K == 401:
        Xh = oracle.z_scores(po.read_csv(os.path.join(DATA, "nir.csv"))); Yh = oracle.z_scores(po.read_csv(os.path.join(DATA, "octane.csv")))
    else:
        Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, 1)
    dtype = np.float64 if dt == "f64" else np.float32
    if dt == "f32":
        Xh = np.asfortranarray(Xh.astype(np.float32).astype(np.float64)); Yh = np.asfortranarray(Yh.astype(np.float32).astype(np.float64))
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    Xbig = torch.zeros((K, N + pad), dtype=torch.float64 if dt == "f64" else torch.float32, device="cuda")
    Xbig[:, :N] = torch.from_numpy(np.ascontiguousarray(Xh.T.astype(dtype))).cuda()
    Xd = Xbig.T[:N]                                   # column-major view, ld = N + pad
    Yd = to_dev(Yh.astype(dtype))
    handle.set_option(pls_amd.OPT_PROFILE, 2)
    try:
        handle.timing()
        out = handle.fit_device(Xd, Yd, A)
        torch.cuda.synchronize()
        t = handle.timing()
        assert sum(t["launches"].values()) == 1, t["launches"]
        with handle_with_env(PLS_HIP_TINY=0) as general:  # (the switches are read when a handle is created)
            general.set_option(pls_amd.OPT_PROFILE, 2)
            plain = general.fit_device(Xd, Yd, A)
            torch.cuda.synchronize()
            assert sum(general.timing()["launches"].values()) > 1
            plain = {k: v.clone() for k, v in plain.items()}
    finally:
        handle.set_option(pls_amd.OPT_PROFILE, 0)
    tol = dict(tol_b=1e-10, tol_col=1e-9) if dt == "f64" else dict(tol_b=2e-5, tol_col=2e-4, tol_inv=1e-3)
    check_against(po, out, ref, Bref, Tref=ref["T"], col_err=cerr, **tol)
    for k in "WPQRB":
        assert np.abs(out[k].cpu().numpy() - plain[k].cpu().numpy()).max() < (1e-9 if dt == "f64" else 1e-4)


@pytest.mark.parametrize("N,K,M,A,dt,pad", [(10, 15, 2, 2, "f64", 0), (60, 40, 4, 6, "f64", 3), (64, 300, 8, 5, "f64", 0), (130, 100, 3, 7, "f64", 1),
                                            (1000, 20, 2, 6, "f32", 8), (700, 26, 8, 4, "f64", 0), (5, 7, 2, 3, "f64", 0)])
def test_single_launch_fit_several_responses(handle, oracle, po, N, K, M, A, dt, pad):
    """2 .. 8 responses as ONE launch (tiny_fit_m_kernel, MM = 2, 4, 8): BASELINE config 1 -- the reference's README example,
    toyX / toyY with two responses and two components -- and synthetic shapes of every row-block layout; against the oracle at
    1e-10 and against the general plan (PLS_HIP_TINY=0)."""
    import pls_amd
    torch = _torch()
    if (N, K, M) == (10, 15, 2):
        Xh = oracle.z_scores(po.read_csv(os.path.join(DATA, "toyX.csv"))); Yh = oracle.z_scores(po.read_csv(os.path.join(DATA, "toyY.csv")))
    else:
        Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    dtype = np.float64 if dt == "f64" else np.float32
    if dt == "f32":
        Xh = np.asfortranarray(Xh.astype(np.float32).astype(np.float64)); Yh = np.asfortranarray(Yh.astype(np.float32).astype(np.float64))
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    Xbig = torch.zeros((K, N + pad), dtype=torch.float64 if dt == "f64" else torch.float32, device="cuda")
    Xbig[:, :N] = torch.from_numpy(np.ascontiguousarray(Xh.T.astype(dtype))).cuda()
    Xd = Xbig.T[:N]
    Yd = to_dev(Yh.astype(dtype))
    handle.set_option(pls_amd.OPT_PROFILE, 2)
    try:
        handle.timing()
        out = handle.fit_device(Xd, Yd, A)
        torch.cuda.synchronize()
        t = handle.timing()
        assert sum(t["launches"].values()) == 1, t["launches"]
        with handle_with_env(PLS_HIP_TINY=0) as general:  # (the switches are read when a handle is created)
            general.set_option(pls_amd.OPT_PROFILE, 2)
            plain = general.fit_device(Xd, Yd, A)
            torch.cuda.synchronize()
            assert sum(general.timing()["launches"].values()) > 1
            plain = {k: v.clone() for k, v in plain.items()}
    finally:
        handle.set_option(pls_amd.OPT_PROFILE, 0)
    tol = dict(tol_b=1e-10, tol_col=1e-9) if dt == "f64" else dict(tol_b=2e-5, tol_col=2e-4, tol_inv=1e-3)
    check_against(po, out, ref, Bref, Tref=ref["T"], col_err=cerr, **tol)
    for k in "WPQRB":
        assert np.abs(out[k].cpu().numpy() - plain[k].cpu().numpy()).max() < (1e-9 if dt == "f64" else 1e-4)


@pytest.mark.parametrize("case", ["toy-loo", "toy-lso", "synth-m4"])
def test_single_launch_folds_several_responses(handle, oracle, po, monkeypatch, case):
    """Cross-validation folds of small multi-response data on the fold form of the single-launch kernel (one workgroup per fold:
    cv_LOO / cv_LSO of the reference's own example, src/pls.cpp:469-549), against one oracle refit per fold and against the
    general form of the call."""
    if case.startswith("toy"):
        Xh = oracle.z_scores(po.read_csv(os.path.join(DATA, "toyX.csv"))); Yh = oracle.z_scores(po.read_csv(os.path.join(DATA, "toyY.csv")))
        A = 2
    else:
        Xh, Yh, A = oracle.synth_x(0, 90, 30), oracle.synth_y(0, 90, 4), 5
    N = Xh.shape[0]
    rng = np.random.default_rng(5)
    idx = np.arange(N)[:, None] if case == "toy-loo" else np.stack([rng.permutation(N)[:max(1, (3 * N) // 10)] for _ in range(12)])
    Xd, Yd = to_dev(Xh), to_dev(Yh)
    got = handle.cv_folds(Xd, Yd, A, idx).cpu().numpy()
    ref = _fold_reference(oracle, Xh, Yh, A, idx)
    assert np.abs(got - ref).max() < 1e-9 * max(np.abs(ref).max(), 1.0)
    with handle_with_env(PLS_HIP_TINY=0) as h0:
        general = h0.cv_folds(Xd, Yd, A, idx).cpu().numpy()
    assert np.abs(got - general).max() < 1e-9 * max(np.abs(ref).max(), 1.0)


def test_graph_replay_of_repeated_fits(oracle, po):
    """PLS_HIP_OPT_GRAPH: the second identical device-memory fit is captured, later ones are one hipGraphLaunch.  Results are
    the eager ones bit for bit; different inputs at the same addresses are picked up (the graph holds pointers, not data);
    a changed shape falls back to capture again."""
    import pls_amd
    torch = _torch()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side), handle_with_env(PLS_HIP_RESIDENT=0) as h:  # (the launches of the general plan, eager and replayed:
        N, K, M, A = 3000, 96, 2, 5                                          #  the one-launch resident fit is never captured)
        X = h.synth_x(0, N, K, 11); Y = h.synth_y(0, N, M, 11)
        for algo in (pls_amd.ALGO_KERNEL, pls_amd.ALGO_NIPALS):
            h.set_option(pls_amd.OPT_ALGO, algo)
            h.set_option(pls_amd.OPT_GRAPH, 0)
            eager = h.fit_device(X, Y, A); side.synchronize()
            want = {k: eager[k].clone() for k in "WPQRTB"}
            h.set_option(pls_amd.OPT_GRAPH, 1)
            o = h.fit_device(X, Y, A)
            for rep in range(4):                      # eager, capture + launch, replay, replay
                for k in "WPQRTB": o[k].zero_()
                h.fit_device(X, Y, A, out=o); side.synchronize()
                for k in "WPQRTB": assert torch.equal(o[k], want[k]), (algo, rep, k)
            X2 = h.synth_x(0, N, K, 12)
            X.copy_(X2)                               # new data, same addresses: the replayed graph fits it
            h.fit_device(X, Y, A, out=o); side.synchronize()
            ref = oracle.plsr(X.cpu().numpy(), Y.cpu().numpy(), A)
            assert po.rel_fro(o["B"].cpu().numpy(), oracle.coefficients(ref["R"], ref["Q"])) < 1e-10
            X.copy_(h.synth_x(0, N, K, 11))
        o3 = h.fit_device(X[:2000], Y[:2000], A); side.synchronize()   # another shape: eager again
        ref = oracle.plsr(X[:2000].cpu().numpy(), Y[:2000].cpu().numpy(), A)
        assert po.rel_fro(o3["B"].cpu().numpy(), oracle.coefficients(ref["R"], ref["Q"])) < 1e-10
        h.set_option(pls_amd.OPT_GRAPH, 0)


def _fold_reference(oracle, Xh, Yh, A, idx):
    nf, ts = idx.shape
    ref = np.zeros((Yh.shape[1], nf * ts, A))
    N = Xh.shape[0]
    for f in range(nf):
        train = np.setdiff1d(np.arange(N), idx[f])
        c = oracle.plsr(Xh[train], Yh[train], A)
        for nc in range(1, A + 1):
            ref[:, f * ts:(f + 1) * ts, nc - 1] = (Yh[idx[f]] - Xh[idx[f]] @ oracle.coefficients(c["R"], c["Q"], nc)).T
    return ref


@pytest.mark.parametrize("N,K,M,A,ts,nf,dt", [(60, 24, 1, 6, 1, 60, "f64"), (200, 24, 3, 5, 60, 7, "f64"), (301, 33, 2, 4, 17, 5, "f32")])
def test_cv_folds_refit_per_fold(handle, oracle, po, monkeypatch, N, K, M, A, ts, nf, dt):
    """The general form of pls_hip_cv_folds (one device refit per fold, the reference's own procedure,
    src/pls.cpp:478-488,:524-545) against the oracle and against the batched launch on the same folds."""
    torch = _torch()
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    if dt == "f32":
        Xh = np.asfortranarray(Xh.astype(np.float32).astype(np.float64)); Yh = np.asfortranarray(Yh.astype(np.float32).astype(np.float64))
    dtype = np.float64 if dt == "f64" else np.float32
    rng = np.random.default_rng(11)
    idx = np.arange(N)[:, None] if ts == 1 and nf == N else np.stack([rng.permutation(N)[:ts] for _ in range(nf)])
    Xd, Yd = to_dev(Xh.astype(dtype)), to_dev(Yh.astype(dtype))
    batched = handle.cv_folds(Xd, Yd, A, idx).cpu().numpy()
    with handle_with_env(PLS_HIP_CV_REFIT=1) as hr:
        refit = hr.cv_folds(Xd, Yd, A, idx).cpu().numpy()
        refit_host = hr.cv_folds(np.asfortranarray(Xh.astype(dtype)), np.asfortranarray(Yh.astype(dtype)), A, idx)
    ref = _fold_reference(oracle, Xh, Yh, A, idx)
    tol = (1e-8 if dt == "f64" else 2e-4) * max(np.abs(ref).max(), 1.0)   # fp32: the refit stores its scores in fp32
    assert np.abs(refit - ref).max() < tol
    assert np.abs(refit_host - refit).max() < 1e-12
    assert np.abs(refit - batched).max() < tol


def test_cv_folds_many_responses(handle, oracle, po):
    """M = 40 responses: beyond the batched fold kernel (M <= 32), served by one refit per fold."""
    N, K, M, A, ts, nf = 150, 50, 40, 3, 30, 4
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    rng = np.random.default_rng(5)
    idx = np.stack([rng.permutation(N)[:ts] for _ in range(nf)])
    E = handle.cv_folds(to_dev(Xh), to_dev(Yh), A, idx).cpu().numpy()
    ref = _fold_reference(oracle, Xh, Yh, A, idx)
    assert np.abs(E - ref).max() < 1e-8 * max(np.abs(ref).max(), 1.0)


def test_batched_cv_folds_fp32_storage(handle, oracle, po):
    torch = _torch()
    N, K, M, A, ts, nf = 400, 40, 2, 4, 25, 9
    X = handle.synth_x(0, N, K, 5, dtype=torch.float32); Y = handle.synth_y(0, N, M, 5, dtype=torch.float32)
    Xh = X.cpu().numpy().astype(np.float64); Yh = Y.cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(3)
    idx = np.stack([rng.permutation(N)[:ts] for _ in range(nf)])
    E = handle.cv_folds(X, Y, A, idx).cpu().numpy()
    Eh = handle.cv_folds(X.cpu().numpy(), Y.cpu().numpy(), A, idx)
    for f in (0, nf - 1):
        train = np.setdiff1d(np.arange(N), idx[f])
        c = oracle.plsr(Xh[train], Yh[train], A)
        for nc in range(1, A + 1):
            want = (Yh[idx[f]] - Xh[idx[f]] @ oracle.coefficients(c["R"], c["Q"], nc)).T
            assert np.abs(E[:, f * ts:(f + 1) * ts, nc - 1] - want).max() < 1e-8     # fp64 accumulation throughout
    assert np.abs(E - Eh).max() < 1e-12
