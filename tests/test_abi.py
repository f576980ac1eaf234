"""CPU checks of the drop-in boundary: the library loads, exports every symbol include/pls_hip.h
declares (and nothing is declared that is not bound in Python), fails loudly without a GPU, and the
host-side partition logic is right.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "pls_hip.h")).read()
    return sorted(set(re.findall(r"PLS_HIP_API\s+(?:const\s+char\s*\*|int)\s*(pls_hip_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import pls_amd
    from pls_amd import _lib
    names = _declared()
    assert len(names) >= 18
    L = pls_amd.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/pls_hip.h but not exported"
    bound = sorted(n for n, _, _ in _lib.PROTOTYPES)
    assert bound == names, "pls_amd/_lib.py PROTOTYPES and include/pls_hip.h disagree"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r" T (pls_hip_\w+)", out)))
    assert exported == names, "the .so exports symbols the header does not declare (or vice versa)"
    assert L.pls_hip_abi_version() == 1


def test_rccl_helper_exports():
    from pls_amd import _lib
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libpls_hip_rccl.so")
    assert os.path.exists(path), "optional RCCL reducer helper not built"
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    src = open(os.path.join(ROOT, "include", "pls_hip_rccl.h")).read()
    declared = sorted(set(re.findall(r"PLS_HIP_API\s+int\s+(pls_hip_rccl_\w+)\s*\(", src)))
    assert sorted(set(re.findall(r" T (pls_hip_rccl_\w+)", out))) == declared and len(declared) == 4


def test_header_is_plain_c():
    """the boundary must compile as C (no C++/torch types in the signatures)"""
    src = '#include "pls_hip.h"\n#include "pls_hip_rccl.h"\nint main(void){return pls_hip_abi_version()==PLS_HIP_ABI_VERSION?0:1;}\n'
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        "-x", "c", "-"], input=src, text=True, capture_output=True)
    assert r.returncode == 0, r.stderr


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pls_amd
    with pytest.raises(pls_amd.PlsHipError) as e:
        pls_amd.Handle()
    assert e.value.code == 2  # PLS_HIP_ERR_DEVICE: no CPU fallback exists
    # NULL handle is rejected, not dereferenced
    assert pls_amd.lib().pls_hip_synchronize(None) == 1
    assert pls_amd.lib().pls_hip_last_error(None) == b"null handle"


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under pls_amd/ or include/ may reference it."""
    bad = []
    for base in ("pls_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".c")):
                    for line in open(os.path.join(dp, f), errors="replace"):
                        if re.search(r"(#\s*include|\bimport\b|\bfrom\b|CDLL|dlopen|LoadLibrary).*oracle", line):
                            bad.append((os.path.join(dp, f), line.strip()))
    assert not bad, bad


def test_row_partition():
    from pls_amd.distributed import row_partition
    for n in (0, 1, 7, 8, 1000, 1 << 20, 16777216 + 3):
        for w in (1, 2, 3, 8):
            blocks = [row_partition(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == n
            for (a0, an), (b0, _) in zip(blocks, blocks[1:]):
                assert a0 + an == b0
            assert max(b[1] for b in blocks) - min(b[1] for b in blocks) <= 1
    with pytest.raises(ValueError):
        row_partition(10, 2, 2)


REF_MAIN = "/root/reference/src/main.cpp"


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference tree is not on this machine (GPU box)")
def test_reference_main_builds_unchanged_against_this_header(tmp_path):
    """The strongest boundary check available without the reference's Eigen: its own CSV-driven main
    (src/main.cpp:1-44), UNCHANGED, compiles against include/PLS/pls.h and links against libpls.so -- every
    type, free function, enum value, constructor and member it uses exists here with a compatible signature."""
    host = os.path.join(ROOT, "pls_amd", "host")
    csrc = os.path.join(ROOT, "pls_amd", "csrc")
    if not os.path.exists(os.path.join(host, "libpls.so")):
        pytest.skip("libpls.so not built")
    exe = str(tmp_path / "ref_main")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), REF_MAIN, "-o", exe,
                        "-L", host, "-lpls", "-L", csrc, "-lpls_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                        f"-Wl,-rpath,{host}", f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # same usage behaviour as the reference (exit(100) on a wrong argument count, src/main.cpp:12-16); no GPU needed
    u = subprocess.run([exe], capture_output=True, text=True)
    assert u.returncode == 100
