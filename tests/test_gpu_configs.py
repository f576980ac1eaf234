"""BASELINE.json configs 3, 4 and 5 at their OWN K, M, A (and, where the card holds it, their own N).

  * twins: N reduced to 4,096 rows, everything else as the config says -- every execution plan against the oracle,
    whose B must reproduce the committed fixture (tests/golden/c?twin_*.npz);
  * full size: the HIP path against the oracle run on the SAME full-size inputs on the host cores (configs 3
    and 4: a 4.3 GB fp64 matrix, 41 / 101 passes -- seconds with the OpenMP build of the oracle), and, for one
    shard of config 5 (17.2 GB), through size-independent properties plus row blocks against the oracle.

Tolerances: fp64 storage B <= 1e-10 (north star); fp32 storage (config 4, no reference counterpart: float_type is
double, include/PLS/pls.h:22) B <= 2e-5 against the fp64 oracle on the fp32-rounded inputs.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from test_gpu_parity import TOL_B, check_against, to_dev

pytestmark = pytest.mark.gpu

_CACHE = {}


def _torch():
    import torch
    return torch


@pytest.fixture(scope="module")
def oracle_omp(po):
    return po.OracleLib(omp=True)


def _twin(name, oracle, po):
    """inputs + live oracle results of a config twin (computed once per session)"""
    if name in _CACHE:
        return _CACHE[name]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    N, K, M, A, seed, f32 = (int(g[k]) for k in ("N", "K", "M", "A", "seed", "f32"))
    Xh, Yh = oracle.synth_x(0, N, K, seed), oracle.synth_y(0, N, M, seed)
    if f32:
        Xh = np.asfortranarray(Xh.astype(np.float32).astype(np.float64))
        Yh = np.asfortranarray(Yh.astype(np.float32).astype(np.float64))
    assert float(np.abs(Xh).sum()) == float(g["x_checksum"]) and float(np.abs(Yh).sum()) == float(g["y_checksum"])
    ref = oracle.plsr(Xh, Yh, A)
    Bref = oracle.coefficients(ref["R"], ref["Q"])
    # the live oracle is the one the fixture was made from
    assert po.rel_fro(Bref, g["B"]) < 1e-13 and np.allclose((ref["T"] ** 2).sum(0), g["tt"], rtol=1e-12)
    _CACHE[name] = (g, Xh, Yh, ref, Bref)
    return _CACHE[name]


@pytest.fixture(params=[(0, 0), (0, 1), (1, 0), (1, 1), (2, 1)],
                ids=["kernel-unfused", "kernel-fused", "nipals-unfused", "nipals-fused", "gram"])
def plan(request, handle):
    import pls_amd
    algo, fuse = request.param
    handle.set_option(pls_amd.OPT_ALGO, algo)
    handle.set_option(pls_amd.OPT_FUSE, fuse)
    yield request.param
    handle.set_option(pls_amd.OPT_ALGO, 0)
    handle.set_option(pls_amd.OPT_FUSE, 1)


def _orthogonality(T):
    """max |cos| between two different score columns"""
    torch = _torch()
    Td = T.to(torch.float64)
    G = (Td.t() @ Td).cpu().numpy()
    d = np.sqrt(np.diag(G))
    C = G / np.outer(d, d)
    return float(np.abs(C - np.eye(len(d))).max())


# ------------------------------------------------------------------------------------------
# config 4: K = 4,096, M = 8, A = 50, fp32 storage
# ------------------------------------------------------------------------------------------
def test_config4_twin_A50_fp32(handle, oracle, po, plan):
    """50 components on fp32 storage: the risk SURVEY section 7 names is the orthogonality of late scores when T
    is rounded to fp32 between the passes.  Every plan against the fp64 oracle on the fp32-rounded inputs."""
    torch = _torch()
    g, Xh, Yh, ref, Bref = _twin("c4twin_4096x4096_m8_A50_f32", oracle, po)
    N, K, M, A = (int(g[k]) for k in ("N", "K", "M", "A"))
    X = handle.synth_x(0, N, K, int(g["seed"]), dtype=torch.float32)
    Y = handle.synth_y(0, N, M, int(g["seed"]), dtype=torch.float32)
    assert np.array_equal(X.cpu().numpy().astype(np.float64), Xh)
    out = handle.fit_device(X, Y, A); handle.synchronize()
    check_against(po, out, ref, Bref, ref["T"], tol_b=2e-5, tol_col=2e-5, col_err=g["col_err"], tol_inv=1e-4)
    assert _orthogonality(out["T"]) < 1e-4           # all 50 scores
    assert np.allclose((out["T"].double().cpu().numpy() ** 2).sum(0), g["tt"], rtol=1e-4)


def test_config4_full_size_A50_fp32(handle, oracle_omp, po):
    """BASELINE config 4 as quoted: 131,072 x 4,096 fp32, m = 8, A = 50.  KERNEL and NIPALS plans against the
    oracle on the same inputs at full size, all 50 scores orthogonal, P^T R = I."""
    import pls_amd
    torch = _torch()
    N, K, M, A = 131072, 4096, 8, 50
    X = handle.synth_x(0, N, K, pls_amd.SEED_DEFAULT, dtype=torch.float32)
    Y = handle.synth_y(0, N, M, pls_amd.SEED_DEFAULT, dtype=torch.float32)
    outs = {}
    for algo in (0, 1):
        handle.set_option(pls_amd.OPT_ALGO, algo)
        outs[algo] = {k: v.clone() for k, v in handle.fit_device(X, Y, A).items()}
        handle.synchronize()
    handle.set_option(pls_amd.OPT_ALGO, 0)
    Xh = np.asfortranarray(X.cpu().numpy().astype(np.float64)); Yh = np.asfortranarray(Y.cpu().numpy().astype(np.float64))
    ref = oracle_omp.plsr(Xh, Yh, A, compensated=True)
    Bref = oracle_omp.coefficients(ref["R"], ref["Q"])
    # per-component conditioning UNDER fp32 STORAGE: how far the storage rounding of the scores (and of the deflated
    # matrix) moves each component, from two independent CPU routes that emulate it (tests/golden/make_golden.py)
    k32 = oracle_omp.plsr(Xh, Yh, A, f32_storage=True)
    n32 = oracle_omp.plsr(Xh, Yh, A, nipals=True, f32_storage=True)
    cerr = np.maximum(po.column_errors(ref, k32), po.column_errors(k32, n32))
    assert po.rel_fro(oracle_omp.coefficients(k32["R"], k32["Q"]), Bref) < 2e-5   # the claim itself, on the CPU
    del Xh, k32, n32
    for algo, o in outs.items():
        check_against(po, o, ref, Bref, None, tol_b=2e-5, tol_col=2e-5, col_err=cerr, tol_inv=1e-4)
        assert _orthogonality(o["T"]) < 1e-4, algo
        tt = (o["T"].double() ** 2).sum(0).cpu().numpy()
        assert np.allclose(tt, (ref["T"] ** 2).sum(0), rtol=1e-4), algo
        s = po.sign_align(ref["W"], o["W"].cpu().numpy())
        for r0 in (0, 54321, N - 700):      # scattered row blocks of several score columns against the oracle's T
            for a in (0, 9, 24, 49):
                got = o["T"][r0:r0 + 700, a].double().cpu().numpy() * s[a]
                want = ref["T"][r0:r0 + 700, a]
                assert np.linalg.norm(got - want) <= max(2e-5, 20 * cerr[a]) * np.linalg.norm(ref["T"][:, a]) / np.sqrt(N / 700), (algo, r0, a)


# ------------------------------------------------------------------------------------------
# config 3: 1,048,576 x 512, m = 1, A = 20, fp64 -- the headline
# ------------------------------------------------------------------------------------------
def test_config3_full_size_A20(handle, oracle_omp, oracle, po):
    """The bench headline as quoted, A = 20, every plan against the oracle on the same full-size inputs:
    B <= 1e-10, W/P/Q/R per component, three scattered row blocks of several score columns, 20 orthogonal scores."""
    import pls_amd
    N, K, M, A = 1 << 20, 512, 1, 20
    X = handle.synth_x(0, N, K, pls_amd.SEED_DEFAULT); Y = handle.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
    Xh = oracle_omp.synth_x(0, N, K); Yh = oracle_omp.synth_y(0, N, M)
    for r0 in (0, 777777, N - 100):   # device and host generators agree at full size
        assert np.array_equal(X[r0:r0 + 100].cpu().numpy(), Xh[r0:r0 + 100])
    # Over 2^20 rows the index-order sums of the plain restatement carry up to N*eps/2 ~ 1e-10 per product -- the size
    # of the tolerance (measured on this matrix: plain vs compensated oracle differ by 1e-9 in B).  The full-size
    # reference therefore runs the same operation sequence with error-free sums (oracle/pls_oracle.c, "Compensated
    # arithmetic"); the plain restatement is required to agree with it to its own summation error only.
    ref = oracle_omp.plsr(Xh, Yh, A, compensated=True)
    Bref = oracle_omp.coefficients(ref["R"], ref["Q"])
    alt = oracle_omp.plsr(Xh, Yh, A, nipals=True, compensated=True)
    cerr = po.column_errors(ref, alt)
    assert cerr.max() < 1e-9, "every one of the 20 components is well determined on this matrix"
    # (the LITERAL restatement: one thread, every sum in index order over all 2^20 rows -- liboracle.so, ~8 s)
    plain = oracle.plsr(Xh, Yh, A)
    Bplain = oracle.coefficients(plain["R"], plain["Q"])
    three = {"plain_vs_compensated_oracle": po.rel_fro(Bplain, Bref)}
    assert three["plain_vs_compensated_oracle"] < 1e-8
    del Xh, alt, plain
    for algo, fuse in ((0, 1), (1, 1), (2, 1), (1, 0)):
        handle.set_option(pls_amd.OPT_ALGO, algo); handle.set_option(pls_amd.OPT_FUSE, fuse)
        try:
            out = handle.fit_device(X, Y, A); handle.synchronize()
        finally:
            handle.set_option(pls_amd.OPT_ALGO, 0); handle.set_option(pls_amd.OPT_FUSE, 1)
        # The three figures behind "B within 1e-10 at full size" (BASELINE.md section 6): the HIP path against the reference
        # ALGORITHM with error-free sums (the bar), against its literal index-order sums (which are themselves ~1e-9 from
        # the error-free ones over 2^20 rows), and those two CPU forms against each other.
        name = {(0, 1): "kernel_fused", (1, 1): "nipals_fused", (2, 1): "gram", (1, 0): "nipals_unfused"}[(algo, fuse)]
        Bhip = out["B"].cpu().numpy()
        three[name] = {"hip_vs_compensated_oracle": po.rel_fro(Bhip, Bref), "hip_vs_plain_oracle": po.rel_fro(Bhip, Bplain)}
        assert three[name]["hip_vs_compensated_oracle"] < 1e-10 and three[name]["hip_vs_plain_oracle"] < 1e-8, three
        check_against(po, out, ref, Bref, None, col_err=cerr)
        assert _orthogonality(out["T"]) < 1e-9, (algo, fuse)
        s = po.sign_align(ref["W"], out["W"].cpu().numpy())
        for r0 in (0, 123456, 600000, N - 2048):
            for a in (0, 5, 11, 19):
                got = out["T"][r0:r0 + 2048, a].cpu().numpy() * s[a]
                want = ref["T"][r0:r0 + 2048, a]
                lim = max(1e-9, 20 * cerr[a]) * np.linalg.norm(ref["T"][:, a]) / np.sqrt(N / 2048)
                assert np.linalg.norm(got - want) <= lim, (algo, fuse, r0, a)
    print("config 3 at full size, relative Frobenius error of B:", three)
    try:  # (the figures BASELINE.md quotes; scratch output of the GPU box)
        import json
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(three, open(os.path.join(ROOT, "gpurun_out", "c3_full_size_parity_three_figures.json"), "w"), indent=1)
    except OSError:
        pass


# ------------------------------------------------------------------------------------------
# config 5: one rank's shard, 2,097,152 x 1,024, m = 4, A = 20, fp64 (bench.py workload "C5rank")
# ------------------------------------------------------------------------------------------
def test_config5_twin_A20(handle, oracle, po, plan):
    g, Xh, Yh, ref, Bref = _twin("c5twin_4096x1024_m4_A20", oracle, po)
    N, K, M, A = (int(g[k]) for k in ("N", "K", "M", "A"))
    X = handle.synth_x(0, N, K, int(g["seed"])); Y = handle.synth_y(0, N, M, int(g["seed"]))
    out = handle.fit_device(X, Y, A); handle.synchronize()
    check_against(po, out, ref, Bref, ref["T"], col_err=g["col_err"])
    assert _orthogonality(out["T"]) < 1e-9


def test_config5_one_shard_full_size(handle, oracle, po):
    """The 17.2 GB shard one GPU owns in config 5 (plus the 17.2 GB working copy of the NIPALS plan): the plans
    agree to 1e-10 on B, scores orthogonal, P^T R = I, row blocks of T against the oracle's product on those rows,
    and the deflation leaves X - t p^T orthogonal to t."""
    import pls_amd
    torch = _torch()
    N, K, M, A = 2097152, 1024, 4, 20
    row0 = 3 * N                       # the shard of rank 3: global row indices beyond 2^22
    X = handle.synth_x(row0, N, K, pls_amd.SEED_DEFAULT); Y = handle.synth_y(row0, N, M, pls_amd.SEED_DEFAULT)
    outs = {}
    for algo in (0, 1):
        handle.set_option(pls_amd.OPT_ALGO, algo)
        try:
            outs[algo] = {k: v.clone() for k, v in handle.fit_device(X, Y, A).items()}
            handle.synchronize()
        finally:
            handle.set_option(pls_amd.OPT_ALGO, 0)
    Bk = outs[0]["B"].cpu().numpy()
    assert np.isfinite(Bk).all()
    assert po.rel_fro(outs[1]["B"].cpu().numpy(), Bk) < TOL_B
    for algo, o in outs.items():
        W = o["W"].cpu().numpy(); P = o["P"].cpu().numpy(); R = o["R"].cpu().numpy()
        assert np.allclose((W * W).sum(0), 1.0, atol=1e-12)
        assert np.allclose(P.T @ R, np.eye(A), atol=1e-9)
        assert _orthogonality(o["T"]) < 1e-9, algo
        for r0 in (0, 1234567, N - 512):
            Xs = oracle.synth_x(row0 + r0, 512, K)
            assert np.array_equal(X[r0:r0 + 512].cpu().numpy(), Xs)
            want = oracle.xb(Xs, R)               # the oracle's X R on those rows
            got = o["T"][r0:r0 + 512].cpu().numpy()
            if algo == 0:
                assert po.rel_fro(got, want) < 1e-11, (algo, r0)
            else:  # NIPALS scores come from the deflated matrix: same values up to the rounding of 20 deflations
                assert po.rel_fro(got, want) < 1e-9, (algo, r0)
    # the two plans' scores agree column by column
    s = po.sign_align(outs[0]["W"].cpu().numpy(), outs[1]["W"].cpu().numpy())
    for a in (0, 10, 19):
        d = float((outs[1]["T"][:, a] * s[a] - outs[0]["T"][:, a]).norm() / outs[0]["T"][:, a].norm())
        assert d < 1e-8, (a, d)
    del outs[1]
    torch.cuda.empty_cache()


def test_config5_at_its_own_size_on_one_gpu(po):
    """BASELINE config 5 -- 16,777,216 x 1,024 fp64, m = 4, A = 20, row-sharded over 8 -- at its OWN row count: eight
    virtual members of 2,097,152 rows each on the one GPU (137.4 GB of X resident in 288 GB of HBM, generated on the device
    by every member at its own row offset).  The real row count, the real number of collectives, 64-bit offsets end to end;
    the only thing missing against an 8-GPU node is the xGMI hop.  KERNEL plan (NIPALS would need a second 137 GB).
    Size-independent checks: the members bit-identical (inside pls_hip_group_fit), P^T R = I, the 20 scores orthogonal
    over all 16.7 M rows, B equal to the GRAM plan's (an independent route: X^T X on the matrix cores), row blocks of T on
    three members against the host product of the same generator rows."""
    import time
    import pls_amd
    torch = pytest.importorskip("torch")
    free, total = torch.cuda.mem_get_info()
    if free < 170e9:
        pytest.skip("needs 170 GB of free HBM")
    N, K, M, A, n = 1 << 24, 1024, 4, 20, 8
    g = pls_amd.Group([0] * n)
    try:
        X = g.synth(N, K, pls_amd.SEED_DEFAULT, "x"); Y = g.synth(N, M, pls_amd.SEED_DEFAULT, "y")
        assert [b[1] for b in g.blocks(X)] == [N // n] * n
        g.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_KERNEL)
        out = g.fit(X, Y, A)                     # warm-up: workspaces
        g.free(out["T"])
        t0 = time.perf_counter(); out = g.fit(X, Y, A); dt = time.perf_counter() - t0
        print(f"config 5 at full size, 8 virtual members on one GPU, KERNEL plan: {dt*1e3:.1f} ms per fit = {A/dt:.1f} components/s "
              f"({g.exchange} exchange)")
        W, P, Q, R, B = (out[k] for k in "WPQRB")
        assert np.isfinite(B).all()
        assert np.allclose((W ** 2).sum(0), 1.0, atol=1e-12)
        assert np.allclose(P.T @ R, np.eye(A), atol=1e-8)
        G = g.gram(out["T"], out["T"])           # T^T T over all the shards
        assert np.abs(G - np.diag(np.diag(G))).max() < 1e-9 * np.diag(G).max()
        # scores of three members against the generator's rows on the host (the last 512 rows of each block)
        class _Raw:  # zero-copy view of device memory the group owns
            def __init__(self, p, k):
                self.__cuda_array_interface__ = {"shape": (k,), "typestr": "<f8", "data": (p, False), "version": 2}

        rows = 512
        for r in (0, 3, 7):
            ptr, ld, r0, nr = g.block(out["T"], r)
            Xh = po.synth_x(r0 + nr - rows, rows, K)
            for c in (0, 7, 19):
                t = torch.as_tensor(_Raw(ptr.value + 8 * (c * ld + nr - rows), rows), device="cuda").cpu().numpy()
                assert po.rel_fro(t, Xh @ R[:, c]) < 1e-10, (r, c)
        g.free(out["T"])
        # the GRAM plan: X^T X of every shard on the matrix cores, summed over the members, component loop on K x K
        g.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_GRAM)
        out2 = g.fit(X, Y, A)
        assert po.rel_fro(out2["B"], B) < 1e-10
        g.free(out2["T"])
    finally:
        g.close()
