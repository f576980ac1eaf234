"""pls_hip_group: one process, several member handles, one host thread per member -- the route PLS::Model takes
(include/PLS/pls.h:187-199) when PLS_HIP_DEVICES asks for more than one GPU.  The test box has ONE GPU, so the members
are virtual shards on device 0: same code path (row partition, resident blocks, per-member streams and threads,
the in-process fixed-order all-reduce over peer pointers), the peer loads just stay on one device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def groups():
    import pls_amd
    made = {}

    def get(n):
        if n not in made:
            made[n] = pls_amd.Group([0] * n)
        return made[n]

    yield get
    for g in made.values():
        g.close()


@pytest.mark.parametrize("n", [1, 2, 3, 8])
@pytest.mark.parametrize("N,K,M,A,algo", [(4096, 64, 1, 6, 0), (4098, 96, 3, 7, 1), (10, 15, 2, 2, 0), (3001, 1300, 2, 5, 1),
                                          (3000, 130, 2, 6, 2)])
def test_group_fit_matches_oracle(groups, oracle, po, n, N, K, M, A, algo):
    """Row-sharded over n members in one process: B <= 1e-10 against the oracle's unsharded fit, scores gathered from
    the members' blocks, and -- checked inside pls_hip_group_fit itself -- every member derived identical bits."""
    import pls_amd
    g = groups(n)
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    ref = oracle.plsr(Xh, Yh, A)
    alt = oracle.plsr(Xh, Yh, A, nipals=True)
    lim = np.maximum(1e-9, 20 * po.column_errors(ref, alt))
    g.set_option(pls_amd.OPT_ALGO, algo)
    try:
        X, Y = g.upload(Xh), g.upload(Yh)
        blocks = g.blocks(X)
        assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == N and all(b[1] % 4 == 0 for b in blocks[:-1])
        assert np.array_equal(g.download(X), Xh)               # the staging pipeline in both directions
        out = g.fit(X, Y, A)
        T = g.download(out["T"])
        assert po.rel_fro(out["B"], oracle.coefficients(ref["R"], ref["Q"])) < 1e-10
        got = dict(out); got["T"] = T
        assert (po.column_errors(ref, got) <= lim).all()
        # predict and metrics through the group: X B on resident data, SSE for every component count
        fv = g.xb(X, out["B"])
        assert po.rel_fro(g.download(fv), Xh @ out["B"]) < 1e-12
        sse = g.model_sse(X, Y, out["R"], out["Q"])
        for c in (1, A):
            assert np.allclose(sse[:, c - 1], po.explained_variance(Xh, Yh, out["R"], out["Q"], c)[1], rtol=1e-8)
        for m in (X, Y, out["T"], fv):
            g.free(m)
    finally:
        g.set_option(pls_amd.OPT_ALGO, 0)


def test_group_members_bit_identical_to_each_other_and_across_group_sizes(groups, oracle):
    """W of an n-member fit equals W of the torch.distributed-style sharded fit only to rounding (different partial
    sums), but within ONE group every member must hold the same bits -- pls_hip_group_fit returns REDUCER otherwise --
    and repeating the fit reproduces them."""
    g = groups(3)
    Xh, Yh = oracle.synth_x(0, 5000, 200), oracle.synth_y(0, 5000, 2)
    X, Y = g.upload(Xh), g.upload(Yh)
    a = g.fit(X, Y, 8)
    b = g.fit(X, Y, 8)
    for k in "WPQRB":
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(g.download(a["T"]), g.download(b["T"]))
    for m in (X, Y, a["T"], b["T"]):
        g.free(m)


def test_group_kernel_type2_and_fp32(groups, oracle, po):
    import pls_amd
    g = groups(2)
    N, K, M, A = 3000, 130, 2, 6
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    ref = oracle.plsr(Xh, Yh, A)
    Bref = oracle.coefficients(ref["R"], ref["Q"])
    X, Y = g.upload(Xh), g.upload(Yh)
    out = g.fit(X, Y, A, method=pls_amd.KERNEL_TYPE2)     # X^T X partial blocks through the in-process reducer
    assert out["T"] is None and po.rel_fro(out["B"], Bref) < 1e-10
    g.free(X); g.free(Y)
    X32, Y32 = g.upload(Xh, np.float32), g.upload(Yh, np.float32)
    o32 = g.fit(X32, Y32, A)
    r32 = oracle.plsr(Xh.astype(np.float32).astype(np.float64), Yh.astype(np.float32).astype(np.float64), A)
    assert po.rel_fro(o32["B"], oracle.coefficients(r32["R"], r32["Q"])) < 2e-5
    assert g.download(o32["T"]).dtype == np.float32
    for m in (X32, Y32, o32["T"]):
        g.free(m)


def test_group_cv_folds_and_errors(groups, oracle, po):
    import pls_amd
    g1, g2 = groups(1), groups(2)
    N, K, M, A = 200, 24, 3, 5
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    X, Y = g1.upload(Xh), g1.upload(Yh)
    idx = np.arange(N)[:, None]
    E = g1.cv_folds(X, Y, A, idx)
    c = oracle.plsr(Xh[1:], Yh[1:], A)
    for nc in range(1, A + 1):
        want = Yh[0] - Xh[0] @ oracle.coefficients(c["R"], c["Q"], nc)
        assert np.abs(E[:, 0, nc - 1] - want).max() < 1e-8
    with pytest.raises(pls_amd.PlsHipError) as e:
        g1.fit(X, Y, K + 1)                                   # A > K
    assert e.value.code == 1
    X2, Y2 = g2.upload(Xh), g2.upload(Yh)
    with pytest.raises(pls_amd.PlsHipError) as e:
        g2.cv_folds(X2, Y2, A, idx)                           # folds need the whole matrix on one device
    assert e.value.code == 4
    with pytest.raises(pls_amd.PlsHipError):
        g2.fit(X, Y2, A)                                      # X belongs to another group's partition (1 block vs 2)
    out = g2.fit(X2, Y2, A)                                   # the group is healthy after the errors
    assert np.isfinite(out["B"]).all()
    for gg, ms in ((g1, (X, Y)), (g2, (X2, Y2, out["T"]))):
        for m in ms:
            gg.free(m)


def test_group_upload_download_large_and_strided(groups, oracle):
    """the pinned staging pipeline: a matrix of several tiles, odd row count, both directions, column ranges"""
    g = groups(3)
    N, K = 700001, 24                       # 134 MB: five 32 MB tiles per member and a ragged tail
    Xh = oracle.synth_x(0, N, K)
    X = g.upload(Xh)
    assert np.array_equal(g.download(X), Xh)
    assert np.array_equal(g.download(X, 5, 3), Xh[:, 5:8])
    g.free(X)


@pytest.mark.parametrize("n", [1, 3])
@pytest.mark.parametrize("N,K,M,A,dt", [(70000, 96, 2, 8, "f64"), (40001, 200, 1, 6, "f64"), (9000, 1300, 3, 5, "f32"),
                                        (10, 15, 2, 2, "f64"), (300000, 64, 1, 5, "f64")])
def test_upload_xy_streamed_gram(groups, oracle, po, n, N, K, M, A, dt):
    """pls_hip_group_upload_xy: X^T X and X^T Y are accumulated row block by row block on the matrix cores while the
    blocks cross PCIe (several 32 MB blocks here, a ragged last one, an odd row count, fp32 storage, empty members) and a
    fit under ALGO_AUTO runs its component loop from them -- same B as the oracle, T = X R, and the same answer as the
    plain upload + KERNEL plan.  Leave-one-out folds take the products from the pair as well."""
    import pls_amd
    g = groups(n)
    dtype = np.float64 if dt == "f64" else np.float32
    Xh, Yh = oracle.synth_x(0, N, K), oracle.synth_y(0, N, M)
    if dt == "f32":
        Xh = np.asfortranarray(Xh.astype(np.float32).astype(np.float64)); Yh = np.asfortranarray(Yh.astype(np.float32).astype(np.float64))
    ref = oracle.plsr(Xh, Yh, A)
    Bref = oracle.coefficients(ref["R"], ref["Q"])
    tol = 1e-10 if dt == "f64" else 2e-5
    g.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_AUTO)
    try:
        X, Y = g.upload_xy(Xh, Yh, dtype)
        assert np.array_equal(g.download(X).astype(np.float64), Xh) and np.array_equal(g.download(Y).astype(np.float64), Yh)
        out = g.fit(X, Y, A)
        assert po.rel_fro(out["B"], Bref) < tol
        T = g.download(out["T"]).astype(np.float64)
        assert po.rel_fro(T, Xh @ out["R"]) < (1e-11 if dt == "f64" else 1e-6)
        o2 = g.fit(X, Y, A, method=pls_amd.KERNEL_TYPE2)          # METHOD::KERNEL_TYPE2 from the same products
        assert po.rel_fro(o2["B"], Bref) < tol
        g.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_KERNEL)       # an explicit plan ignores them
        o3 = g.fit(X, Y, A)
        assert po.rel_fro(o3["B"], out["B"]) < tol
        if n == 1 and N <= 100:
            E = g.cv_folds(X, Y, A, np.arange(N)[:, None])
            c = oracle.plsr(Xh[1:], Yh[1:], A)
            assert np.abs(E[:, 0, A - 1] - (Yh[0] - Xh[0] @ oracle.coefficients(c["R"], c["Q"]))).max() < 1e-8
        for m in (X, Y, out["T"], o3["T"]):
            g.free(m)
    finally:
        g.set_option(pls_amd.OPT_ALGO, 0)


def test_device_side_exchange_virtual_members():
    """The device-side exchange (every member pushes its partial sums into its peers' inboxes and spins on sequence flags:
    no host thread, event or copy per collective) is the default when every member has its own GPU.  Virtual members share
    one GPU and wait for each other's kernels there, which takes one hardware queue per member: a process of its own with
    GPU_MAX_HW_QUEUES set (tools/group_exchange_check.py: plans x shapes x 2, 3, 4, 8 members against single-handle fits)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PLS_HIP_GROUP_EXCHANGE="device", GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "group_exchange_check.py"), "2", "3", "4", "8"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "exchange check ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_device_side_exchange_time_out_and_recovery():
    """A member that never delivers (fault injection: one push dropped) must not hang the others: every member's wait ends at
    the time limit, the call returns PLS_HIP_ERR_REDUCER, the group re-synchronises its sequence numbers, and the next fit
    is right again."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import pls_amd
from oracle import pls_oracle as po
ora = po.OracleLib()
g = pls_amd.Group([0, 0, 0])
assert g.exchange == "device"
N, K, M, A = 3000, 40, 2, 5
Xh, Yh = ora.synth_x(0, N, K), ora.synth_y(0, N, M)
ref = ora.plsr(Xh, Yh, A); Bref = ora.coefficients(ref["R"], ref["Q"])
X, Y = g.upload(Xh), g.upload(Yh)
try:
    g.fit(X, Y, A)            # member 1 drops the push of its collective 4 (self-test = 1, X^T Y = 2, components from 3)
    print("no error")
except pls_amd.PlsHipError as e:
    print("error", e.code, str(e)[:120])
out = g.fit(X, Y, A)          # sequence numbers start over; the injected collective number does not come again
print("refit", po.rel_fro(out["B"], Bref) < 1e-10)
''' % root
    env = dict(os.environ, PLS_HIP_GROUP_EXCHANGE="device", GPU_MAX_HW_QUEUES="16", PLS_HIP_XCHG_TIMEOUT_S="1.5",
               PLS_HIP_TEST_DROP_PUSH="1:4", PLS_AMD_LIBRARY=os.path.join(root, "pls_amd", "csrc", "testing", "libpls_hip.so"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "error 5" in r.stdout and "refit True" in r.stdout, r.stdout[-1500:]
