"""Every matrix shape the reference accepts runs on the one-sweep (tile-resident) kernels: odd row counts, row counts
that are not a multiple of 4 in fp32 storage, columns that are not 16-byte aligned (ld = N odd, an offset base pointer),
leading dimensions beyond 2^31 / 32 bytes.  The reference's products (src/pls.cpp:419-421) work for any row count and
any Eigen map; here those shapes used to fall back to the one-product kernels at twice the traffic.

Each case checks the values against the oracle on the same inputs AND the plan taken (HIP-event families of the fit).
"""
import numpy as np
import pytest

from test_gpu_parity import check_against, oracle_ref

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _place(Xh, dt, ld_extra, base_off):
    """column-major device copy of Xh with ld = rows + ld_extra, starting base_off elements into its allocation"""
    torch = _torch()
    n, k = Xh.shape
    ld = n + ld_extra
    flat = torch.full((k * ld + base_off + 8,), float("nan"), dtype=dt, device="cuda")  # NaN wherever the matrix is not
    v = flat[base_off:base_off + k * ld].view(k, ld)[:, :n].t()
    v.copy_(torch.from_numpy(Xh).to(dt))
    return v, flat


SHAPES = [
    # N, K, M, A, dtype, ld_extra, base_off
    (1001, 37, 2, 6, "f64", 0, 0),      # odd N, ld = N: every second column at 8 mod 16   (4 columns per lane)
    (1001, 37, 2, 6, "f64", 0, 1),      # ... and the base pointer at 8 mod 16
    (1001, 37, 2, 6, "f64", 1, 0),      # odd N in 16-byte aligned columns (ld = N + 1)
    (1001, 37, 2, 3, "f64", 0, 0),      # too few components for the copy into tiles to pay: unaligned columns read directly
    (1000, 200, 1, 7, "f64", 1, 0),     # even N, odd ld                                    (8 columns per lane)
    (4099, 400, 1, 6, "f64", 0, 0),     # odd N, several tiles per workgroup                (16 columns per lane)
    (2051, 513, 3, 5, "f64", 0, 1),     # K > 512: half-height tiles of the working copy    (32 columns per lane)
    (3, 5, 1, 2, "f64", 0, 0),          # fewer rows than a tile; one row pack and a half
    (1, 4, 1, 1, "f64", 0, 0),          # a single row
    (33, 40, 2, 5, "f64", 0, 1),        # one full tile + one row
    (1001, 37, 2, 6, "f32", 0, 0),      # fp32: N % 4 = 1
    (1001, 37, 2, 6, "f32", 3, 0),      # fp32: N % 4 = 1 in aligned columns
    (1002, 130, 1, 6, "f32", 0, 1),     # fp32: N % 4 = 2, base at 4 mod 16
    (2051, 300, 2, 5, "f32", 0, 3),     # fp32: N % 4 = 3, base at 12 mod 16
    (4100, 600, 1, 5, "f32", 1, 0),     # fp32: N % 4 = 0 but ld odd
    (515, 1500, 2, 5, "f64", 0, 0),     # wide: short tiles of 8 fp64 rows (128 column groups)
    (515, 3000, 1, 5, "f64", 0, 1),     # wide: 4-row tiles (256 column groups)
    (1031, 2500, 2, 4, "f32", 0, 1),    # wide fp32
    (261, 5000, 1, 4, "f64", 0, 0),     # beyond 4096 columns: row-pack tiles of the copy (512 column groups)
    (131, 8192, 2, 5, "f64", 0, 1),     # ... the widest they take, unaligned source
    (263, 6000, 8, 4, "f32", 0, 0),
    (265, 5000, 12, 4, "f64", 0, 0),    # ... with more than 8 responses: both plans still run fused on the copy
    (130, 4100, 3, 3, "f32", 1, 2),
    (75, 9000, 1, 4, "f64", 0, 0),      # 8192 < K <= 16384 on a short matrix: the one-product kernels
    (4099, 8200, 1, 3, "f64", 0, 0),    # ... and from 4096 rows on the KERNEL plan still fuses (32 columns per lane, read-only)
    (67, 16384, 2, 3, "f32", 0, 1),
    (41, 17000, 1, 3, "f64", 0, 0),     # beyond every resident tile: the one-product kernels
    (512, 20000, 1, 3, "f64", 0, 0),    # ... short and wide: the score kernel splits the columns as well as the rows
    (301, 40000, 2, 3, "f32", 0, 1),
    (2100, 18000, 1, 2, "f64", 1, 0),   # (16-byte accesses from 8 row groups on)
    (100, 20000, 5, 4, "f64", 0, 0),    # several responses beyond the cooperative update's 16,384 columns: three launches
    (64, 17000, 8, 3, "f64", 0, 1),
    (90, 33000, 3, 5, "f32", 0, 0),
    (4100, 9000, 1, 3, "f32", 0, 0),    # NIPALS beyond 8192 columns / KERNEL fused
    (1, 4097, 1, 1, "f64", 0, 0),       # the smallest cases of the wide paths: one row, one component
    (2, 8193, 2, 1, "f64", 0, 1),
    (3, 16385, 1, 2, "f64", 0, 0),
    (5, 8192, 2, 3, "f32", 0, 1),
    (6, 16384, 1, 4, "f32", 0, 0),
    # odd N with every tile height of the working copy and enough components for an error in one row's contribution to
    # show (a lane just behind the swept rows once picked up t_prev of the tail row: 1e-5 in P from the second component on)
    (1365, 1024, 4, 9, "f64", 0, 0),    # 16-row tiles (512 < K <= 1024)
    (1365, 520, 4, 8, "f64", 1, 0),
    (1365, 128, 2, 10, "f64", 0, 0),    # 128-row tiles of narrow matrices (8 column groups)
    (1381, 200, 2, 10, "f64", 0, 1),    # 64-row tiles (16 column groups)
    (1381, 700, 1, 12, "f64", 0, 0),
    (1367, 900, 2, 8, "f32", 0, 0),
    (1365, 100, 1, 8, "f32", 0, 2),
    (4099, 30, 1, 6, "f64", 0, 0),      # 4 columns per lane of the 128-row tile
    # the copy + X^T Y sweep on the tall source tiles, several responses (Y block through LDS, two tiles in flight)
    (1001, 100, 8, 6, "f32", 0, 1),
    (1002, 200, 5, 6, "f64", 1, 0),
    (2049, 64, 3, 5, "f32", 0, 0),
    (2050, 120, 4, 12, "f64", 0, 0),    # aligned columns, enough components for the KERNEL plan's copy
    (2052, 250, 8, 12, "f32", 0, 0),
]


@pytest.fixture(scope="module")
def handle():
    """(this module's handle: small single-response fits on the GENERAL plan -- PLS_HIP_TINY=0, read when the handle is
    created -- they would otherwise run as one launch and never reach the kernels under test)"""
    from conftest import handle_with_env
    torch = _torch()
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    with handle_with_env(PLS_HIP_TINY=0) as h:
        yield h


@pytest.fixture(params=[(0, 1), (1, 1)], ids=["kernel", "nipals"])
def plan(request, handle):
    import pls_amd
    handle.set_option(pls_amd.OPT_ALGO, request.param[0])
    handle.set_option(pls_amd.OPT_FUSE, 1)
    handle.set_option(pls_amd.OPT_PROFILE, 1)
    yield request.param
    handle.set_option(pls_amd.OPT_ALGO, 0)
    handle.set_option(pls_amd.OPT_PROFILE, 0)


@pytest.mark.parametrize("N,K,M,A,dt,ld_extra,base_off", SHAPES)
def test_edge_shapes_take_the_one_sweep_plan(handle, oracle, po, plan, monkeypatch, N, K, M, A, dt, ld_extra, base_off):
    import pls_amd
    torch = _torch()
    tdt = torch.float64 if dt == "f64" else torch.float32
    Xh = oracle.synth_x(0, N, K); Yh = oracle.synth_y(0, N, M)
    if dt == "f32":
        Xh = Xh.astype(np.float32).astype(np.float64); Yh = Yh.astype(np.float32).astype(np.float64)
    X, keepx = _place(Xh, tdt, ld_extra, base_off)
    Y, keepy = _place(Yh, tdt, ld_extra, base_off)
    # T with ld = N as well, between guard values
    ldt = N + ld_extra
    tflat = torch.full((A * ldt + base_off + 8,), 7.0, dtype=tdt, device="cuda")
    T = tflat[base_off:base_off + A * ldt].view(A, ldt)[:, :N].t()
    out = {k: pls_amd.colmajor_empty(K, A, torch.float64, "cuda", ld=K) for k in "WPR"}
    out["Q"] = pls_amd.colmajor_empty(M, A, torch.float64, "cuda", ld=M)
    out["B"] = pls_amd.colmajor_empty(K, M, torch.float64, "cuda", ld=K)
    out["T"] = T
    handle.timing()
    handle.fit_device(X, Y, A, out=out); handle.synchronize()
    tm = handle.timing()
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    if dt == "f64":
        check_against(po, out, ref, Bref, ref["T"], col_err=cerr)
    else:
        check_against(po, out, ref, Bref, ref["T"], tol_b=2e-5, tol_col=2e-5, col_err=cerr, tol_inv=1e-4)
    # nothing outside the N x A scores was written, nothing of the caller's X either
    guard = tflat.clone()
    guard[base_off:base_off + A * ldt].view(A, ldt)[:, :N] = 7.0
    assert bool((guard == 7.0).all()), "score stores outside the N x A matrix"
    assert np.array_equal(X.cpu().numpy().astype(np.float64), Xh)
    # the plan taken: the one-sweep kernels, not the one-product ones
    nipals = plan[0] == 1
    V = 2 if dt == "f64" else 4
    unaligned = (N + ld_extra) % V != 0 or base_off % V != 0
    if K <= 1024:
        assert tm["launches"]["fused"] == A and tm["launches"]["xb"] == 0, tm["launches"]
        if not nipals and ((unaligned and A >= 4) or (not unaligned and A >= 10 and K <= 512)):
            # KERNEL plan on unaligned columns: one copy into aligned tiles, formed in the same sweep as X^T Y
            assert tm["launches"]["deflate"] == 1 and tm["launches"]["xty"] == 0, tm["launches"]
        else:
            assert tm["launches"]["deflate"] == 0, tm["launches"]
    elif A < 3:
        pass   # (the copy into short tiles is taken from the third component on; values checked above)
    elif K <= 4096:
        # short tiles: one copy into them (in the same sweep as X^T Y) + A fused passes, read-only (KERNEL) or in place (NIPALS)
        assert tm["launches"]["fused"] == A and tm["launches"]["deflate"] == 1 and tm["launches"]["xty"] == 0, tm["launches"]
    elif K <= 8192:
        # row-pack tiles: the same plan (one copy in the X^T Y sweep + A fused passes; more than 8 responses: a plain copy and
        # the X^T Y pass of its own)
        assert tm["launches"]["fused"] == A and tm["launches"]["deflate"] == 1, tm["launches"]
        assert (tm["launches"]["xty"] == 0) if M <= 8 else (tm["launches"]["xty"] >= 1), tm["launches"]
    elif K <= 16384 and not nipals and N >= 4096:
        # 32 columns per lane, read-only: from 4096 rows on (on a shorter matrix the partial rows of K doubles per workgroup
        # weigh as much as the matrix, and the one-product kernels are the faster plan)
        assert tm["launches"]["fused"] == A and tm["launches"]["deflate"] == 1 and tm["launches"]["xty"] == 0, tm["launches"]
    else:
        assert tm["launches"]["fused"] == 0, tm["launches"]


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_edge_results_equal_the_aligned_ones(handle, po, dt):
    """the same values in an aligned and in an unaligned layout: the component loop runs the same arithmetic (the EDGE
    instantiations only change how a tile reaches the registers); the prologue's X^T Y sums in another order"""
    import pls_amd
    torch = _torch()
    tdt = torch.float64 if dt == "f64" else torch.float32
    N, K, M, A = 4096 + 64, 300, 2, 6
    Xa = handle.synth_x(0, N, K, 77, dtype=tdt); Ya = handle.synth_y(0, N, M, 77, dtype=tdt)
    Xu, keep = _place(Xa.cpu().numpy(), tdt, 1, 1)
    for algo in (0, 1):
        handle.set_option(pls_amd.OPT_ALGO, algo)
        a = handle.fit_device(Xa, Ya, A); handle.synchronize()
        b = handle.fit_device(Xu, Ya, A); handle.synchronize()
        assert po.rel_fro(b["B"].cpu().numpy(), a["B"].cpu().numpy()) < (1e-12 if dt == "f64" else 1e-6), algo
    handle.set_option(pls_amd.OPT_ALGO, 0)


def test_very_tall_matrix_runs_fused(handle, po):
    """one matrix of 2^24 rows x 64 fp64 columns (8.6 GB): 32 column groups of it span 4.3 GB, more than one buffer
    descriptor addresses -- the EDGE instantiations build a descriptor per wave (4 columns) and keep the one-sweep plan"""
    import pls_amd
    N, K, M, A = 1 << 24, 64, 1, 4
    X = handle.synth_x(0, N, K, 3); Y = handle.synth_y(0, N, M, 3)
    outs = {}
    handle.set_option(pls_amd.OPT_PROFILE, 1)
    for algo in (0, 1):
        handle.set_option(pls_amd.OPT_ALGO, algo)
        handle.timing()
        outs[algo] = handle.fit_device(X, Y, A); handle.synchronize()
        tm = handle.timing()
        assert tm["launches"]["fused"] == A and tm["launches"]["xb"] == 0, tm["launches"]
    handle.set_option(pls_amd.OPT_FUSE, 0)
    handle.set_option(pls_amd.OPT_ALGO, 0)
    outs["unfused"] = handle.fit_device(X, Y, A); handle.synchronize()
    handle.set_option(pls_amd.OPT_FUSE, 1)
    handle.set_option(pls_amd.OPT_PROFILE, 0)
    Bb = outs["unfused"]["B"].cpu().numpy()
    for k in (0, 1):
        assert po.rel_fro(outs[k]["B"].cpu().numpy(), Bb) < 1e-10, k
    T = outs[1]["T"]; R = outs[1]["R"].cpu().numpy()
    for r0 in (0, (1 << 23) + 12345, N - 1000):
        rows = slice(r0, r0 + 1000)
        assert po.rel_fro(T[rows, 1].cpu().numpy(), X[rows].cpu().numpy() @ R[:, 1]) < 1e-11


@pytest.mark.parametrize("N,K,dt,ld_extra,base_off", [
    (1001, 37, "f64", 0, 0), (1001, 37, "f64", 0, 1), (4099, 20, "f64", 1, 0), (70001, 17, "f64", 0, 1),
    (1001, 37, "f32", 0, 0), (1002, 40, "f32", 0, 1), (2051, 33, "f32", 0, 3), (70003, 5, "f32", 1, 2), (3, 5, "f64", 0, 1),
])
def test_z_scores_on_unaligned_layouts(handle, oracle, N, K, dt, ld_extra, base_off):
    """column statistics and z-scores (src/pls.cpp:69-111) of matrices whose columns are not 16-byte aligned: the
    element-wise instantiations of the one-sweep statistics and of the scale pass, out of place into an unaligned Z too"""
    torch = _torch()
    tdt = torch.float64 if dt == "f64" else torch.float32
    rng = np.random.default_rng(N * 7 + K)
    Xh = oracle.synth_x(0, N, K) * rng.uniform(0.1, 30, K) + rng.uniform(-50, 50, K)
    if dt == "f32":
        Xh = Xh.astype(np.float32).astype(np.float64)
    X, keepx = _place(Xh, tdt, ld_extra, base_off)
    Z, mean, sd = handle.colwise_z_scores(X); handle.synchronize()
    xl = Xh.astype(np.longdouble)
    mr = xl.mean(0); sr = np.sqrt(((xl - mr) ** 2).sum(0) / (N - 1))
    assert np.allclose(mean.cpu().numpy(), mr.astype(np.float64), rtol=1e-13, atol=1e-13)
    assert np.allclose(sd.cpu().numpy(), sr.astype(np.float64), rtol=1e-12)
    zr = ((xl - mr) / sr).astype(np.float64)
    assert np.abs(Z.cpu().numpy() - zr).max() < (1e-10 if dt == "f64" else 5e-6)
    # in place on the unaligned matrix; nothing outside it is touched (the NaN guard cells stay NaN, the matrix is finite)
    Z2, _, _ = handle.colwise_z_scores(X, inplace=True); handle.synchronize()
    assert np.abs(X.cpu().numpy() - zr).max() < (1e-10 if dt == "f64" else 5e-6)
    n_nan = int(torch.isnan(keepx).sum())
    assert n_nan == keepx.numel() - N * K


@pytest.mark.parametrize("N,K,C,dt,ld_extra,base_off", [
    (512, 20000, 1, "f64", 0, 0), (300, 9000, 3, "f64", 0, 1), (1000, 5000, 4, "f64", 1, 0), (77, 33000, 7, "f64", 0, 0),
    (2100, 3000, 2, "f32", 0, 0), (515, 12000, 5, "f32", 0, 3), (40000, 1500, 3, "f64", 0, 0), (9, 70000, 2, "f64", 0, 0), (130, 6000, 45, "f64", 0, 0),
])
def test_scores_of_short_wide_matrices(handle, oracle, N, K, C, dt, ld_extra, base_off):
    """X * B (scores and fitted values, src/pls.cpp:439-451) of matrices with few rows and many columns: the columns
    are split over workgroups as well as the rows (xb_split_kernel), several columns of B per sweep"""
    torch = _torch()
    tdt = torch.float64 if dt == "f64" else torch.float32
    rng = np.random.default_rng(N + K + C)
    Xh = oracle.synth_x(0, N, K)
    if dt == "f32":
        Xh = Xh.astype(np.float32).astype(np.float64)
    Bh = np.asfortranarray(rng.standard_normal((K, C)))
    X, keep = _place(Xh, tdt, ld_extra, base_off)
    out = handle.xb(X, torch.from_numpy(Bh).cuda()); handle.synchronize()
    ref = Xh @ Bh
    err = np.abs(out.cpu().numpy().astype(np.float64) - ref).max() / np.abs(ref).max()
    assert err < (1e-13 if dt == "f64" else 2e-7), err


@pytest.mark.parametrize("N,K,M,A,dt", [(300, 9000, 1, 4, "f64"), (64, 20000, 3, 3, "f64"), (501, 6000, 2, 5, "f32"), (33, 16384, 1, 3, "f64")])
def test_host_memory_entry_on_wide_matrices(handle, oracle, po, N, K, M, A, dt):
    """pls_hip_fit on HOST matrices (the entry PLS::Model uses) beyond 4096 columns: staging, the copy into row-pack tiles /
    the split score kernel, results back in host memory -- against the oracle"""
    Xh, Yh = oracle.synth_x(0, N, K, seed=N + K), oracle.synth_y(0, N, M, seed=N + K)
    npdt = np.float64 if dt == "f64" else np.float32
    if dt == "f32":
        Xh = Xh.astype(np.float32).astype(np.float64); Yh = Yh.astype(np.float32).astype(np.float64)
    ref, Bref, cerr = oracle_ref(oracle, po, Xh, Yh, A)
    out = handle.fit_host(Xh.astype(npdt), Yh.astype(npdt), A, dtype=npdt)
    tol = 1e-10 if dt == "f64" else 2e-5
    assert po.rel_fro(np.asarray(out["B"], dtype=np.float64), Bref) < max(tol, 50 * float(np.max(cerr)))
