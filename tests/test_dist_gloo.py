"""The N>1 path on CPU: two processes over torch.distributed/gloo run the row-sharded algorithm
 with the PRODUCT's partitioning and all-reduce plumbing (pls_amd.distributed) and
the oracle's sharded restatement as the per-rank compute; the result must equal the unsharded fit.
The same sharding on the real HIP path is covered by tests/test_gpu_dist.py (-m gpu)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, K, M, A, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import pls_oracle as po
    from pls_amd.distributed import make_host_allreduce, row_partition
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ora = po.OracleLib()
        row0, nrows = row_partition(N, world, rank)
        X = ora.synth_x(row0, nrows, K); Y = ora.synth_y(row0, nrows, M)
        out = ora.plsr_sharded(X, Y, K, M, A, make_host_allreduce())
        q.put((rank, row0, {k: np.array(v) for k, v in out.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,K,M,A,world", [(1001, 24, 1, 6, 2), (777, 16, 3, 5, 2), (3, 5, 2, 2, 2), (50, 9, 1, 4, 3)])
def test_sharded_equals_unsharded(N, K, M, A, world):
    import torch.multiprocessing as mp
    from oracle import pls_oracle as po
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, K, M, A, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ora = po.OracleLib()
    X = ora.synth_x(0, N, K); Y = ora.synth_y(0, N, M)
    ref = ora.plsr(X, Y, A)
    Bref = ora.coefficients(ref["R"], ref["Q"])
    T = np.concatenate([r[2]["T"] for r in res], axis=0)
    for rank, _, out in res:
        assert po.rel_fro(ora.coefficients(out["R"], out["Q"]), Bref) < 1e-11
        for k in "WPQR":   # replicated outputs are bit-identical on every rank
            assert np.array_equal(out[k], res[0][2][k]), k
    s = po.sign_align(ref["W"], res[0][2]["W"])
    assert po.rel_fro(T * s, ref["T"]) < 1e-9


def test_sharded_column_statistics_recipe():
    """The row-sharded z-score statistics of the device path (colmoments_shard_kernel, src/pls.cpp:69-83) restated in
    numpy: every shard holds (count, mean, M2) of its rows; all-reduce 1 sums n_r * mean_r (-> the global mean g),
    all-reduce 2 sums M2_r + n_r (mean_r - g)^2 (-> SST about g).  Checked against the two-pass statistics of the whole
    matrix on columns with a large offset, with an empty shard and a one-row shard among the ranks."""
    rng = np.random.default_rng(4)
    N, K = 5003, 7
    X = rng.standard_normal((N, K)) * rng.uniform(0.1, 30, K) + np.array([0, 1e6, -3e8, 5, 0.1, 1e3, -40])
    for splits in ([2050, 2044, 909], [2500, 0, 2503], [1, 5001, 1], [N]):
        assert sum(splits) == N
        tri, row0 = [], 0
        for n_r in splits:
            blk = X[row0:row0 + n_r]; row0 += n_r
            # (the device carries a shard's mean as an unevaluated sum hi + lo: the merge is FIRST order in its error, and at an
            # offset of 1e8 sd a plain fp64 mean would cost 1e-10 in the sd -- extended precision stands in for hi + lo here)
            bl = blk.astype(np.longdouble)
            mean_r = bl.mean(0) if n_r else np.zeros(K, dtype=np.longdouble)
            m2_r = (((bl - mean_r) ** 2).sum(0)).astype(np.float64) if n_r else np.zeros(K)
            tri.append((float(n_r), mean_r, m2_r))
        g = (sum(n * m.astype(np.float64) for n, m, _ in tri) / N)               # all-reduce 1 (plain fp64 sums)
        sst = sum(q + n * ((m - g) ** 2).astype(np.float64) for n, m, q in tri)  # all-reduce 2 (second order in the error of g)
        sd = np.sqrt(sst / (N - 1))
        xl = X.astype(np.longdouble)
        ref_mean = xl.mean(0); ref_sd = np.sqrt(((xl - ref_mean) ** 2).sum(0) / (N - 1))
        assert np.allclose(g, ref_mean.astype(np.float64), rtol=1e-13, atol=1e-15)   # (a mean near 0 cancels)
        assert np.allclose(sd, ref_sd.astype(np.float64), rtol=1e-12, atol=0), splits
