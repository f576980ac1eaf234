"""Row-sharded fit on the real HIP path: two (and three) ranks share the one GPU of the test box,
each owning a row block, the small partial products summed through pls_amd.distributed's reducer
(gloo here -- RCCL refuses two ranks on one device; the nccl branch of the same reducer is what
bench.py uses on a multi-GPU node).  The sharded result must match the oracle's unsharded fit and be
bit-identical across ranks."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _split(N, world, rank, splits):
    """(row0, nrows): the even partition of pls_amd.distributed, or explicit block sizes (ragged / empty shards)"""
    if splits is None:
        from pls_amd.distributed import row_partition
        return row_partition(N, world, rank)
    assert sum(splits) == N and len(splits) == world
    return sum(splits[:rank]), splits[rank]


def _worker(rank, world, port, N, K, M, A, algo, fuse, q, method=0, splits=None, reducer="torch", env=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.update(env or {})
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import pls_amd
    from pls_amd.distributed import attach_reducer, row_partition
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        h = pls_amd.Handle()
        h.set_option(pls_amd.OPT_ALGO, algo)
        h.set_option(pls_amd.OPT_FUSE, fuse)
        row0, nrows = _split(N, world, rank, splits)
        X = h.synth_x(row0, nrows, K, pls_amd.SEED_DEFAULT)
        Y = h.synth_y(row0, nrows, M, pls_amd.SEED_DEFAULT)
        if reducer == "ipc":   # the library's device-side exchange between processes (gloo only carries the set-up)
            from pls_amd.distributed import attach_ipc_exchange
            attach_ipc_exchange(h)
        else:
            attach_reducer(h, K, M)
        out = h.fit_device(X, Y, A, method=method)
        h.synchronize()
        out = h.fit_device(X, Y, A, method=method, out=out)   # (a second fit: sequence numbers carried on)
        h.synchronize()
        res = {k: v.cpu().numpy() for k, v in out.items() if v is not None}   # T is None for KERNEL_TYPE2
        # sharded pre-processing and metrics use the same reducer (column statistics over all rows)
        Z, mean, sd = h.colwise_z_scores(X, n_total=N)
        h.synchronize()
        res["mean"] = mean.cpu().numpy(); res["sd"] = sd.cpu().numpy()
        q.put((rank, res))
        h.close()
    except BaseException:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
        raise
    finally:
        dist.destroy_process_group()


def _run(world, N, K, M, A, algo, fuse, method=0, splits=None, timeout=180, reducer="torch", env=None):
    """spawn `world` ranks sharing the GPU; returns [(rank, outputs)] sorted by rank"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, K, M, A, algo, fuse, q, method, splits, reducer, env))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=timeout) for _ in procs), key=lambda t: t[0])
    assert not any("error" in r[1] for r in res), [r[1].get("error") for r in res]
    for p in procs:
        p.join(timeout=timeout)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,splits,algo", [(2, None, 1), (2, None, 0), (3, [2050, 2044, 2], 1), (3, [2048, 0, 2048], 0),
                                               (3, [1365, 1366, 1365], 1)],
                         ids=["2ranks-nipals", "2ranks-kernel", "3ranks-tiny-shard-nipals", "3ranks-empty-shard-kernel",
                              "3ranks-odd-shards-nipals"])
def test_sharded_config5_shape(world, splits, algo):
    """BASELINE config 5's own K = 1,024, m = 4, A = 20 (N reduced to the 4,096 rows of the committed twin), row-sharded
    over 2 and 3 ranks -- even shards, a 2-row shard, an empty shard, odd shard sizes (no 16-byte row packs: the
    one-product kernels on those ranks, the fused pass on the others).  Against the committed fixture and the oracle."""
    from conftest import GOLDEN
    from oracle import pls_oracle as po
    g = np.load(os.path.join(GOLDEN, "c5twin_4096x1024_m4_A20.npz"))
    N, K, M, A = (int(g[k]) for k in ("N", "K", "M", "A"))
    res = _run(world, N, K, M, A, algo, 1, splits=splits)
    ora = po.OracleLib()
    X = ora.synth_x(0, N, K); Y = ora.synth_y(0, N, M)
    ref = ora.plsr(X, Y, A)
    for rank, out in res:
        assert po.rel_fro(out["B"], g["B"]) < 1e-10, rank
        for k in "WPQRB":
            assert np.array_equal(out[k], res[0][1][k]), (rank, k)
    T = np.concatenate([out["T"] for _, out in res], axis=0)
    assert T.shape == (N, A)
    s = po.sign_align(ref["W"], res[0][1]["W"])
    lim = np.maximum(1e-9, 20 * g["col_err"])
    err = np.linalg.norm(T * s - ref["T"], axis=0) / np.linalg.norm(ref["T"], axis=0)
    assert (err <= lim).all(), err
    got = dict(res[0][1]); got["T"] = T
    assert (po.column_errors(ref, got) <= lim).all()


@pytest.mark.parametrize("N,K,M,A,algo,splits", [(600, 6000, 1, 5, 0, [301, 299]), (600, 6000, 1, 5, 1, [300, 0, 300]),
                                                 (400, 20000, 3, 4, 0, [150, 250]), (900, 9000, 1, 4, 1, [450, 450])],
                         ids=["kernel-row-pack-tiles", "nipals-empty-rank", "several-responses-beyond-16384-columns", "nipals-beyond-8192"])
def test_sharded_wide_matrices(N, K, M, A, algo, splits):
    """Row-sharded fits on matrices beyond 4096 columns: the ranks copy their shards into row-pack tiles (or, beyond the
    tiles' reach, take the split score kernel), an odd or empty shard takes another local path than its peers -- the
    sequence of collectives is the same on every rank, the K-sized update (on many workgroups here) runs on identical
    reduced sums: replicas bit-identical, coefficients at the oracle's."""
    from oracle import pls_oracle as po
    res = _run(len(splits), N, K, M, A, algo, 1, splits=splits)
    ora = po.OracleLib()
    X = ora.synth_x(0, N, K); Y = ora.synth_y(0, N, M)
    ref = ora.plsr(X, Y, A)
    Bref = ora.coefficients(ref["R"], ref["Q"])
    for rank, out in res:
        assert po.rel_fro(out["B"], Bref) < 1e-10, rank
        for k in "WPQRB":
            assert np.array_equal(out[k], res[0][1][k]), (rank, k)
    T = np.concatenate([out["T"] for _, out in res], axis=0)
    s = po.sign_align(ref["W"], res[0][1]["W"])
    assert np.abs(T * s - ref["T"]).max() / np.abs(ref["T"]).max() < 1e-8


@pytest.mark.parametrize("algo,method,splits", [(2, 0, [1500, 0, 1500]), (0, 1, [1500, 0, 1500]), (2, 0, [1001, 998, 1001]),
                                                (0, 1, [1001, 998, 1001])],
                         ids=["gram-empty-rank", "type2-empty-rank", "gram-odd-shards", "type2-odd-shards"])
def test_sharded_xx_collective_shape(algo, method, splits):
    """X^T X of a sharded KERNEL_TYPE2 / GRAM fit with K > 32 when the ranks take DIFFERENT local paths: an empty
    rank (memset), odd row counts (the 32-column-block fallback instead of the matrix-core SYRK).  The exchange is one
    all-reduce of 8*K*K values on every rank whatever its local path; a mismatch would hang or corrupt XX."""
    from oracle import pls_oracle as po
    N, K, M, A = 3000, 130, 2, 6
    res = _run(3, N, K, M, A, algo, 1, method=method, splits=splits)
    ora = po.OracleLib()
    X = ora.synth_x(0, N, K); Y = ora.synth_y(0, N, M)
    ref = ora.plsr(X, Y, A)
    Bref = ora.coefficients(ref["R"], ref["Q"])
    for rank, out in res:
        assert po.rel_fro(out["B"], Bref) < 1e-10, rank
        assert np.array_equal(out["W"], res[0][1]["W"])


@pytest.mark.parametrize("N,K,M,A,world,algo,fuse", [
    (4098, 64, 1, 6, 2, 0, 1), (4098, 64, 1, 6, 2, 1, 1), (3001, 40, 3, 5, 2, 1, 0), (2, 5, 1, 2, 3, 0, 1),
    (65536, 512, 1, 4, 3, 1, 1), (2052, 1300, 2, 5, 2, 1, 1), (2052, 1300, 2, 5, 2, 0, 1), (1030, 2500, 1, 4, 2, 1, 1)])
def test_sharded_fit_matches_oracle(N, K, M, A, world, algo, fuse):
    import torch.multiprocessing as mp
    from oracle import pls_oracle as po
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, K, M, A, algo, fuse, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    assert not any("error" in r[1] for r in res), [r[1].get("error") for r in res]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ora = po.OracleLib()
    X = ora.synth_x(0, N, K); Y = ora.synth_y(0, N, M)
    ref = ora.plsr(X, Y, A)
    Bref = ora.coefficients(ref["R"], ref["Q"])
    for rank, out in res:
        assert po.rel_fro(out["B"], Bref) < 1e-10
        for k in "WPQRB":  # every rank derives the same bits from the reduced partials
            assert np.array_equal(out[k], res[0][1][k]), (rank, k)
    T = np.concatenate([out["T"] for _, out in res], axis=0)
    s = po.sign_align(ref["W"], res[0][1]["W"])
    alt = ora.plsr(X, Y, A, nipals=True)
    lim = np.maximum(1e-9, 20 * po.column_errors(ref, alt))
    err = np.linalg.norm(T * s - ref["T"], axis=0) / np.linalg.norm(ref["T"], axis=0)
    assert (err <= lim).all()


@pytest.mark.parametrize("algo,method", [(2, 0), (0, 1)], ids=["gram", "kernel_type2"])
def test_sharded_gram_and_type2(algo, method):
    """the X^T X partial blocks (8*K*K values, a library-owned buffer) go through the same reducer"""
    import torch.multiprocessing as mp
    from oracle import pls_oracle as po
    N, K, M, A, world = 3000, 130, 2, 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, K, M, A, algo, 1, q, method)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    assert not any("error" in r[1] for r in res), [r[1].get("error") for r in res]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ora = po.OracleLib()
    X = ora.synth_x(0, N, K); Y = ora.synth_y(0, N, M)
    ref = ora.plsr(X, Y, A)
    for rank, out in res:
        assert po.rel_fro(out["B"], ora.coefficients(ref["R"], ref["Q"])) < 1e-10
        assert np.array_equal(out["W"], res[0][1]["W"])
        assert np.allclose(out["mean"], X.mean(0), rtol=1e-12, atol=1e-13)
        assert np.allclose(out["sd"], X.std(0, ddof=1), rtol=1e-11)
    if method == 0:
        T = np.concatenate([out["T"] for _, out in res], axis=0)
        s = po.sign_align(ref["W"], res[0][1]["W"])
        assert po.rel_fro(T * s, ref["T"]) < 1e-8


def _nccl_worker(port, q):
    # (PLS_HIP_RESIDENT=0: the reducer-free fit of this size would otherwise be the one-launch resident fit, whose sums meet
    # in another order -- the comparison below is bit for bit between the SAME launches with and without a reducer)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", PLS_HIP_RESIDENT="0")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import pls_amd
    from pls_amd.distributed import attach_reducer
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        N, K, M, A = 8192, 128, 2, 5
        h = pls_amd.Handle()
        X = h.synth_x(0, N, K, 1); Y = h.synth_y(0, N, M, 1)
        plain = h.fit_device(X, Y, A); h.synchronize()
        attach_reducer(h, K, M)          # RCCL all-reduce (a 1-rank communicator) between the kernels
        red = h.fit_device(X, Y, A); h.synchronize()
        q.put({k: bool(torch.equal(plain[k], red[k])) for k in "WPQRTB"})
        h.close()
    finally:
        dist.destroy_process_group()


def test_nccl_reducer_single_rank():
    """the nccl (RCCL) branch of the reducer, stream-ordered between the library's kernels: with one
    rank the sum is the identity, so the fit must be bit-identical to the reducer-free fit."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    same = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert all(same.values()), same


def _rccl_direct_worker(q):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    os.environ["PLS_HIP_RESIDENT"] = "0"  # (the same launches with and without the reducer: see _nccl_worker)
    sys.path.insert(0, ROOT)
    import torch
    import pls_amd
    from pls_amd.distributed import attach_rccl_reducer, detach_rccl_reducer
    try:
        torch.cuda.set_device(0)
        N, K, M, A = 8192, 128, 2, 5
        h = pls_amd.Handle()
        X = h.synth_x(0, N, K, 1); Y = h.synth_y(0, N, M, 1)
        plain = h.fit_device(X, Y, A); h.synchronize()
        attach_rccl_reducer(h)            # ncclCommInitRank(nranks = 1) + ncclAllReduce on the launch stream
        red = h.fit_device(X, Y, A); h.synchronize()
        detach_rccl_reducer(h)
        again = h.fit_device(X, Y, A); h.synchronize()
        q.put({k: bool(torch.equal(plain[k], red[k]) and torch.equal(plain[k], again[k])) for k in "WPQRTB"})
        h.close()
    except BaseException:
        import traceback
        q.put({"error": traceback.format_exc()})
        raise


def test_library_rccl_reducer_single_rank():
    """include/pls_hip_rccl.h: the library-owned RCCL communicator as reducer (the C++ host route)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_direct_worker, args=(q,))
    p.start()
    same = q.get(timeout=120)
    p.join(timeout=60)
    assert "error" not in same, same.get("error")
    assert p.exitcode == 0 and all(same.values()), same


def _guard_worker(rank, world, port, q, corrupt):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import pls_amd
    from pls_amd.distributed import attach_reducer, row_partition
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        N, K, M, A = 4096, 96, 2, 6
        h = pls_amd.Handle()
        h.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_NIPALS)
        row0, nrows = row_partition(N, world, rank)
        X = h.synth_x(row0, nrows, K, pls_amd.SEED_DEFAULT); Y = h.synth_y(row0, nrows, M, pls_amd.SEED_DEFAULT)

        def post(view, call):  # a reducer that leaves DIFFERENT sums on the ranks: rank 1's third collective is off by one ulp-ish
            if corrupt and rank == 1 and call == 2:
                view[0] += 1e-9 * (abs(float(view[0])) + 1.0)

        attach_reducer(h, K, M, post=post)
        h.fit_device(X, Y, A)
        status = "ok"
        try:
            h.synchronize()
        except pls_amd.PlsHipError as e:
            status = f"error {e.code}"
        # the handle stays usable: the next (clean) fit passes the guard again
        attach_reducer(h, K, M)
        h.fit_device(X, Y, A); h.synchronize()
        q.put((rank, {"status": status}))
        h.close()
    except BaseException:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("corrupt", [False, True], ids=["clean", "one-rank-off"])
def test_replica_divergence_guard(corrupt):
    """Every rank of a sharded fit must derive the same bits.  The library checks it itself after the component loop (a
    checksum of W, P, Q, R, B, compared across the ranks with one 512-byte sum all-reduce) and pls_hip_synchronize reports
    PLS_HIP_ERR_REDUCER on EVERY rank when a reducer left different sums on one of them."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, q, corrupt)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=180)
    assert not any("error" in r[1] for r in res), [r[1].get("error") for r in res]
    want = "error 5" if corrupt else "ok"
    assert [r[1]["status"] for r in res] == [want, want], res


@pytest.mark.parametrize("world,splits,algo,method", [(2, None, 1, 0), (3, [2048, 0, 2048], 0, 0), (3, [1365, 1366, 1365], 1, 0),
                                                      (2, None, 0, 1)],
                         ids=["2ranks-nipals", "3ranks-empty-shard-kernel", "3ranks-odd-shards-nipals", "2ranks-type2"])
def test_sharded_fit_over_the_ipc_exchange(world, splits, algo, method):
    """One process per rank, the ranks' partial sums exchanged by the library itself: every rank writes into the other
    ranks' inboxes (hipIpcOpenMemHandle) and spins on sequence flags -- no RCCL, no torch in a collective.  Config 5's own
    K = 1,024, m = 4, A = 20 on the 4,096-row twin; KERNEL_TYPE2 sends its 8 x K x K partial of X^T X in 512 KB pieces.
    Bit-identical ranks, B against the committed fixture."""
    from conftest import GOLDEN
    from oracle import pls_oracle as po
    g = np.load(os.path.join(GOLDEN, "c5twin_4096x1024_m4_A20.npz"))
    N, K, M, A = (int(g[k]) for k in ("N", "K", "M", "A"))
    res = _run(world, N, K, M, A, algo, 1, method=method, splits=splits, reducer="ipc")
    X = po.OracleLib().synth_x(0, N, K)
    for rank, out in res:
        assert po.rel_fro(out["B"], g["B"]) < 1e-10, rank
        for k in "WPQRB":
            assert np.array_equal(out[k], res[0][1][k]), (rank, k)
        # the column statistics of ALL rows went through the same exchange (sliced messages: K % 8 == 0 here, the case in
        # which an unsliced message would have been folded silently)
        assert np.allclose(out["mean"], X.mean(0), rtol=1e-12, atol=1e-13), rank
        assert np.allclose(out["sd"], X.std(0, ddof=1), rtol=1e-11), rank
    if method == 0:
        T = np.concatenate([out["T"] for _, out in res], axis=0)
        G = T.T @ T
        assert np.abs(G - np.diag(np.diag(G))).max() < 1e-9 * np.diag(G).max()


@pytest.mark.parametrize("tail", ["1", "2"], ids=["update-kernel", "update-in-the-tail"])
@pytest.mark.parametrize("splits,algo", [([8192 + 64, 8192 - 64], 1), ([12288, 512, 0, 3584], 1), ([16000, 384], 0)],
                         ids=["2ranks-nipals", "4ranks-small-and-empty-shards-nipals", "2ranks-one-small-kernel"])
def test_sharded_one_response_routes_agree_bit_for_bit(splits, algo, tail):
    """One response over the device-side exchange.  The ranks of one fit may take DIFFERENT routes through a component: a
    shard of 32 tiles or more sums its partial rows in the tail of the pass and pushes from there (PLS_HIP_TAIL=2: it also waits
    for the peers and runs the update there -- one launch per component), a smaller one goes pass -> reduce -> exchange -> update
    kernel, an empty one has no pass at all.  The one-response update is ONE piece of arithmetic for all of them
    (update_m1.hpp), so the replicas must agree bit for bit -- the library's own guard checks it too -- and with the oracle."""
    from oracle import pls_oracle as po
    N, K, M, A = sum(splits), 192, 1, 7
    res = _run(len(splits), N, K, M, A, algo, 1, splits=splits, reducer="ipc", env={"PLS_HIP_TAIL": tail})
    ora = po.OracleLib()
    Xh, Yh = ora.synth_x(0, N, K), ora.synth_y(0, N, M)
    ref = ora.plsr(Xh, Yh, A)
    Bref = ora.coefficients(ref["R"], ref["Q"])
    for rank, out in res:
        assert po.rel_fro(out["B"], Bref) < 1e-10, rank
        for k in "WPQRB":
            assert np.array_equal(out[k], res[0][1][k]), (rank, k)
    T = np.concatenate([out["T"] for _, out in res], axis=0)
    for a in range(A):
        s = np.sign(T[:, a] @ ref["T"][:, a])
        assert po.rel_fro(s * T[:, a], ref["T"][:, a]) < 1e-9, a


def _ipc_timeout_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      PLS_HIP_XCHG_TIMEOUT_S="1.5", PLS_HIP_TEST_DROP_PUSH="1:4",
                      PLS_AMD_LIBRARY=os.path.join(ROOT, "pls_amd", "csrc", "testing", "libpls_hip.so"))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import pls_amd
    from pls_amd.distributed import attach_ipc_exchange, row_partition
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        N, K, M, A = 4096, 64, 1, 6
        h = pls_amd.Handle()
        row0, nrows = row_partition(N, world, rank)
        X = h.synth_x(row0, nrows, K, pls_amd.SEED_DEFAULT); Y = h.synth_y(row0, nrows, M, pls_amd.SEED_DEFAULT)
        attach_ipc_exchange(h)            # (self-test = collective 1)
        h.fit_device(X, Y, A)             # rank 1 drops the push of collective 4
        status = "ok"
        try:
            h.synchronize()
        except pls_amd.PlsHipError as e:
            status = f"error {e.code}"
        q.put((rank, {"status": status}))
        h.close()
    except BaseException:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
        raise
    finally:
        dist.destroy_process_group()


def test_ipc_exchange_time_out_is_reported_on_every_rank():
    """A rank that never delivers a partial sum (fault injection: one push dropped) must not hang the others: every rank's wait
    ends at the time limit and pls_hip_synchronize returns PLS_HIP_ERR_REDUCER on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ipc_timeout_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=180)
    assert not any("error" in r[1] for r in res), [r[1].get("error") for r in res]
    assert [r[1]["status"] for r in res] == ["error 5", "error 5"], res
