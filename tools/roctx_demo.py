"""A small fit under PLS_HIP_ROCTX=1 for `rocprofv3 --marker-trace --kernel-trace`: the ranges the library pushes
(pls_hip_fit, X^T Y, component a, ...) next to the kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pls_amd
h = pls_amd.Handle()
X = h.synth_x(0, 65536, 256, 1); Y = h.synth_y(0, 65536, 2, 1)
for algo in (pls_amd.ALGO_KERNEL, pls_amd.ALGO_NIPALS, pls_amd.ALGO_GRAM):
    h.set_option(pls_amd.OPT_ALGO, algo)
    h.fit_device(X, Y, 4); h.synchronize()
print("done")
