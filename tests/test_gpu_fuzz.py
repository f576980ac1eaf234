"""Randomised parity sweep in the GPU suite: tools/fuzz_parity.py, 60 cases at a fixed seed WITH random handle options
(work-buffer layout, fused / one-product kernels, deferred write-back, fused-pass grid) -- random shapes (single-launch sizes,
small, Gram-sized, wide, very wide), 1-8 responses, both storage types, aligned and unaligned layouts, every plan plus
KERNEL_TYPE2 and random cross-validation fold sets, each against the CPU oracle on the same inputs (B at 1e-10 in fp64)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_random_shapes_plans_and_options_against_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "60", "4", "options"],
                       capture_output=True, text=True, timeout=1200)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0 and "done: 60 cases, 0 bad" in r.stdout, tail
