"""CPU-side sanitizer runs (AddressSanitizer + UndefinedBehaviorSanitizer; GPU sanitizers are not available on
this pool): the oracle's C restatement on every golden-sized path, and the host utilities of the C++ drop-in
(dense matrix family, CSV reader, column statistics, Wilcoxon, normalcdf, shuffles) compiled with the sanitizers
and compared with numpy.  No device call anywhere in this file."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import DATA, ROOT

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def test_oracle_under_asan_ubsan(tmp_path):
    """build liboracle_asan.so and drive every entry point of the oracle through it in a child interpreter
    (libasan must be loaded first): fits of all three forms, fp32-storage and compensated modes, ragged shapes,
    the sharded variant, the generators."""
    lib = str(tmp_path / "liboracle_asan.so")
    subprocess.check_call(["gcc", "-std=c11", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-unknown-pragmas", *SAN,
                           "-o", lib, os.path.join(ROOT, "oracle", "pls_oracle.c"), "-lm"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    code = r'''
import sys, numpy as np, ctypes
sys.path.insert(0, %r)
from oracle import pls_oracle as po
ref = po.OracleLib()                # prototypes of the regular build ...
o = po.OracleLib()
o.lib = L = ctypes.CDLL(%r)         # ... applied to the sanitized one, behind the same wrapper methods
for fn in ("oracle_plsr", "oracle_plsr_nipals", "oracle_plsr_sharded", "oracle_coefficients", "oracle_xb", "oracle_xty",
           "oracle_xv", "oracle_xtv", "oracle_colwise_z_scores", "oracle_dominant_eigvec_sts", "oracle_synth_x",
           "oracle_synth_y", "oracle_num_threads", "oracle_set_f32_storage", "oracle_set_compensated"):
    getattr(L, fn).restype = getattr(ref.lib, fn).restype
    getattr(L, fn).argtypes = getattr(ref.lib, fn).argtypes
for (N, K, M, A) in ((9, 7, 2, 4), (1, 3, 1, 1), (40, 1, 1, 1), (257, 33, 8, 6), (64, 70, 3, 5)):
    X = o.synth_x(5, N, K); Y = o.synth_y(5, N, M)
    a = o.plsr(X, Y, A); b = o.plsr(X, Y, A, nipals=True); c = o.plsr(X, Y, A, method=1)
    d = o.plsr(X, Y, A, compensated=True); e = o.plsr(X, Y, A, f32_storage=True)
    B = o.coefficients(a["R"], a["Q"]); o.coefficients(a["R"], a["Q"], 1)
    assert po.rel_fro(o.coefficients(d["R"], d["Q"]), B) < 1e-9
    o.xb(X, B); o.xty(X, Y); o.z_scores(X); o.dominant_eigvec(o.xty(X, Y))
    if N >= 2:
        o.plsr_sharded(X[: N // 2], Y[: N // 2], K, M, min(A, N // 2), lambda v: None)
print("ok")
''' % (ROOT, lib)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def _parse(text):
    out, name, rows = {}, None, []
    for line in text.splitlines():
        if line.startswith("@"):
            if name:
                out[name] = rows
            parts = line[1:].split()
            name, rows = parts[0], []
            if len(parts) > 1 and not (len(parts) == 3 and parts[1].isdigit() and parts[2].isdigit()):
                out[name] = parts[1:]      # a one-line entry
                name = None
        elif name is not None:
            rows.append(line)
    if name:
        out[name] = rows
    return out


def test_host_utilities_under_asan_ubsan(tmp_path):
    """the C++ host utilities (no device call) compiled with ASan + UBSan against include/ and pls_amd/host/pls.cpp's
    own source, on the reference's nir example: numbers against numpy, printing against Eigen's default format."""
    exe = str(tmp_path / "host_utils")
    csrc = os.path.join(ROOT, "pls_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", *SAN, "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "host_utils.cpp"), os.path.join(ROOT, "pls_amd", "host", "pls.cpp"),
                        "-o", exe, "-L", csrc, "-lpls_hip", "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{csrc}",
                        "-Wl,-rpath,/opt/rocm/lib", "-pthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:protect_shadow_gap=0", UBSAN_OPTIONS="halt_on_error=1")
    p = subprocess.run([exe, os.path.join(DATA, "toyX.csv")], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _parse(p.stdout)
    X = np.loadtxt(os.path.join(DATA, "toyX.csv"), delimiter=",")
    mat = lambda k: np.array([[float(v) for v in row.split()] for row in d[k]])
    assert d["split"] == ["[1.5]", "[]", "[x]", "[]"]
    assert np.array_equal(mat("X"), X)
    assert np.allclose(mat("SST"), ((X - X.mean(0)) ** 2).sum(0)[None, :], rtol=1e-13)
    assert np.allclose(mat("stdev"), X.std(0, ddof=1)[None, :], rtol=1e-13)
    Z = (X - X.mean(0)) / X.std(0, ddof=1)
    assert np.allclose(mat("Z"), Z, rtol=1e-12, atol=1e-13) and np.allclose(mat("zrow"), Z[:1], rtol=1e-12, atol=1e-13)
    ncdf = [float(v) for v in d["normalcdf"]]
    assert abs(ncdf[0] - 0.5) < 1e-3 and abs(ncdf[1] - 0.975) < 2e-3 and abs(ncdf[2] - 0.00135) < 5e-4   # A&S 26.2.19
    w = [float(v) for v in d["wilcoxon"]]
    assert 0.0 <= w[0] <= 1.0 and abs(w[0] + w[1] - 1.0) < 1e-9
    ch = d["choose"]
    bar = ch.index("|")
    assert sorted(int(v) for v in ch[:bar] + ch[bar + 1:]) == list(range(10)) and bar == 7
    assert [int(v) for v in d["ordered"]] == list(np.argsort(X[:, 0], kind="stable"))
    assert d["roundtrip"] == ["1"]
    assert d["print"] == ["    1 -22.5", "  333     4"]              # Eigen default IOFormat: right-aligned to the widest
    assert d["cprint"] == ["(1.5,0)  (-2,0)"]


def test_copy_pool_and_repack_under_tsan(tmp_path):
    """the host threads of the staging pipeline under ThreadSanitizer: the pool's epoch handshake and the repacking of
    strided tiles in both directions (pls_amd/csrc/host_pipeline.hpp); no HIP call is made."""
    exe = str(tmp_path / "copy_pool")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
                        os.path.join(ROOT, "tests", "cpp", "copy_pool.cpp"), "-o", exe, "-L", "/opt/rocm/lib", "-lamdhip64",
                        "-Wl,-rpath,/opt/rocm/lib", "-pthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1", PLS_HIP_STAGE_MB="32"))
    assert p.returncode == 0 and p.stdout.startswith("ok"), (p.stdout[-500:], p.stderr[-3000:])
