"""The C++ drop-in (include/PLS/pls.h + pls_amd/host/pls.cpp) on the GPU: the reference's README
smoke test `PLS toyX.csv toyY.csv 2` and an exerciser of the whole PLS::Model API, compared with the
oracle.  The binaries are built by __graft_entry__.build() and travel with the repo snapshot."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import DATA, GOLDEN, ROOT

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "pls_amd", "host", "PLS")
API = os.path.join(ROOT, "tests", "cpp", "model_api")


def _parse_dump(text):
    out, name, rows = {}, None, []
    for line in text.splitlines():
        if line.startswith("@"):
            if name:
                out[name] = np.array(rows, dtype=np.float64)
            parts = line[1:].split()
            if len(parts) == 3 and parts[1].isdigit():
                name, rows = parts[0], []
            else:
                out[parts[0]] = np.array([float(x) for x in parts[1:]])
                name = None
        elif name is not None and line.strip():
            rows.append([float(x) for x in line.split()])
    if name:
        out[name] = np.array(rows, dtype=np.float64)
    return out


def _oracle_loo(oracle, X, Y, A):
    """cv_LOO restated on the oracle (reference src/pls.cpp:469-491): residual of the left-out row for
    1..A components from a fit on the other N-1 rows."""
    N = X.shape[0]
    E = np.zeros((Y.shape[1], N, A))
    for i in range(N):
        keep = np.arange(N) != i
        c = oracle.plsr(X[keep], Y[keep], A)
        for nc in range(1, A + 1):
            B = oracle.coefficients(c["R"], c["Q"], nc)
            E[:, i, nc - 1] = Y[i] - X[i] @ B
    return E


@pytest.mark.parametrize("devices", [None, "0,0,0"], ids=["one-device", "three-virtual-members"])
@pytest.mark.parametrize("fx,fy,A", [("toyX.csv", "toyY.csv", 2), ("nir.csv", "octane.csv", 4), ("gen:36x5000", "gen:36x2", 4)])
def test_model_api(oracle, po, fx, fy, A, devices, tmp_path):
    """devices = "0,0,0": PLS_HIP_DEVICES spreads the rows of every matrix the Model touches over three members of a
    pls_hip_group (virtual shards on the one GPU of the test box) -- same public API, same numbers.
    gen:NxK: a generated matrix written as CSV (few rows, 5000 columns: the copy into row-pack tiles, the split score
    kernel and the many-workgroup update behind the C++ API)."""
    if fx.startswith("gen:"):
        for name, spec, gen in (("x.csv", fx, oracle.synth_x), ("y.csv", fy, oracle.synth_y)):
            n, k = (int(v) for v in spec[4:].split("x"))
            np.savetxt(tmp_path / name, gen(0, n, k, seed=99), delimiter=",", fmt="%.17g")
        fx, fy = str(tmp_path / "x.csv"), str(tmp_path / "y.csv")
    env = dict(os.environ)
    env.pop("PLS_HIP_DEVICES", None)
    if devices:
        env["PLS_HIP_DEVICES"] = devices
    r = subprocess.run([API, os.path.join(DATA, fx), os.path.join(DATA, fy), str(A)], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _parse_dump(r.stdout)
    X = oracle.z_scores(po.read_csv(os.path.join(DATA, fx)))
    Y = oracle.z_scores(po.read_csv(os.path.join(DATA, fy)))
    assert np.allclose(d["X"], X, rtol=1e-13, atol=1e-14) and np.allclose(d["Y"], Y, rtol=1e-13, atol=1e-14)
    ref = oracle.plsr(X, Y, A)
    B = oracle.coefficients(ref["R"], ref["Q"])
    assert po.rel_fro(d["coefficients"], B) < 1e-10
    assert po.rel_fro(d["coefficients_refit"], B) < 1e-10
    assert po.rel_fro(d["coefficients1"], oracle.coefficients(ref["R"], ref["Q"], 1)) < 1e-10
    s = po.sign_align(ref["P"], d["loadingsX"])
    assert po.rel_fro(d["scores"] * s, ref["T"]) < 1e-9
    assert po.rel_fro(d["loadingsX"] * s, ref["P"]) < 1e-9 and po.rel_fro(d["loadingsY"] * s, ref["Q"]) < 1e-9
    assert po.rel_fro(d["fitted"], X @ B) < 1e-10
    assert po.rel_fro(d["residuals"], Y - X @ B) < 1e-8
    ev, sse = po.explained_variance(X, Y, ref["R"], ref["Q"], A)
    assert np.allclose(d["SSE"].ravel(), sse, rtol=1e-9) and np.allclose(d["EV"].ravel(), ev, rtol=1e-8, atol=1e-10)
    nd = np.stack([Y[:, 0] - (X @ oracle.coefficients(ref["R"], ref["Q"], nc))[:, 0] for nc in range(1, A + 1)], axis=1)
    assert po.rel_fro(d["newdata0"], nd) < 1e-8
    E = _oracle_loo(oracle, X, Y, A)
    assert po.rel_fro(d["loo0"], E[0]) < 1e-7
    assert np.allclose(d["loo_mse"], (E ** 2).mean(axis=1), rtol=1e-7)
    assert d["loo_opt"].shape == (Y.shape[1], 1) and (d["loo_opt"] >= 1).all() and (d["loo_opt"] <= A).all()
    assert d["lso_mse"].shape == (Y.shape[1], A) and np.isfinite(d["lso_mse"]).all()
    assert (d["lso_mse"] > 0.2 * d["loo_mse"]).all() and (d["lso_mse"] < 5 * d["loo_mse"]).all()
    assert d["threw"][0] == 1
    assert abs(d["normalcdf"][0] - 0.6915) < 2e-3 and abs(d["normalcdf"][1] - 0.1056) < 2e-3


def test_readme_smoke_cli(oracle, po):
    """`PLS ../toyX.csv ../toyY.csv 2` (reference README.md:23): everything goes to stderr, matrices
    of (re,im) pairs at 6 significant digits; compare the parsed numbers with the golden fit."""
    r = subprocess.run([CLI, os.path.join(DATA, "toyX.csv"), os.path.join(DATA, "toyY.csv"), "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == ""
    g = np.load(os.path.join(GOLDEN, "toy_A2.npz"))
    txt = r.stderr
    sec = {}
    for name, nxt in (("P", "W"), ("W", "R"), ("R", "Q"), ("Q", "T"), ("T", "coefficients"), ("coefficients", "1 components")):
        body = txt.split(name + ":\n", 1)[1].split(nxt, 1)[0]
        vals = re.findall(r"\(([-+0-9.eE]+),([-+0-9.eE]+)\)", body)
        assert vals and all(float(im) == 0.0 for _, im in vals)
        sec[name] = np.array([float(re_) for re_, _ in vals])
    s = po.sign_align(g["W"], sec["W"].reshape(15, 2))
    for name, shape in (("P", (15, 2)), ("W", (15, 2)), ("R", (15, 2)), ("Q", (2, 2)), ("T", (10, 2))):
        got = sec[name].reshape(shape) * s
        assert np.allclose(got, g[name], rtol=2e-5, atol=2e-6), name   # 6 significant digits printed
    assert np.allclose(sec["coefficients"].reshape(15, 2), g["B"], rtol=2e-5, atol=2e-6)
    m = re.search(r"2 components explained variance:\s+([-0-9.e]+)\s+([-0-9.e]+)\s+- SSE:\s+([-0-9.e]+)\s+([-0-9.e]+)", txt)
    assert m, txt[-1500:]
    assert np.allclose([float(m.group(1)), float(m.group(2))], g["explained_variance"][1], rtol=2e-5)
    assert np.allclose([float(m.group(3)), float(m.group(4))], g["SSE"][1], rtol=2e-5)
    assert "LOO Validation:" in txt and "LSO Validation:" in txt
    assert txt.count("Optimal number of components (by Y variable):") == 2


def test_cli_usage_and_ragged_input(tmp_path):
    r = subprocess.run([CLI], capture_output=True, text=True)
    assert r.returncode == 100 and "Usage" in r.stderr           # reference src/main.cpp:12-16
    bad = tmp_path / "bad.csv"
    bad.write_text("1,2,3\n4,5\n")
    r = subprocess.run([CLI, str(bad), str(bad), "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "columns" in r.stderr           # reference src/pls.cpp:54-58


def test_concurrent_models_and_set_devices():
    """Two host threads build, use and cross-validate their own Models at the same time -- bit-identical to the same work
    done serially (every thread has a device context of its own; nothing is process-global); a Model handed to another
    thread keeps its context; PLS::set_devices moves NEW Models to three virtual members (tests/cpp/concurrent_models.cpp)."""
    exe = os.path.join(ROOT, "tests", "cpp", "concurrent_models")
    env = dict(os.environ)
    env.pop("PLS_HIP_DEVICES", None)
    r = subprocess.run([exe, os.path.join(DATA, "nir.csv"), os.path.join(DATA, "octane.csv")], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "concurrent ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
