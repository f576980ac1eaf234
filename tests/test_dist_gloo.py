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
