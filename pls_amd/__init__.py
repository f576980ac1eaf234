"""pls_amd -- MI355X-native PLS fit / predict hot path behind the tjhladish/PLS API.

Layout: csrc/ (HIP kernels + the C-ABI of include/pls_hip.h), host/ (the C++ drop-in for the
reference's src/pls.cpp and its CSV-driven main), model.py (Python mirror of PLS::Model),
distributed.py (row sharding + torch.distributed reducer).  No CPU fallback anywhere.
"""
from ._lib import (ALGO_AUTO, ALGO_GRAM, ALGO_KERNEL, ALGO_NIPALS, F32, F64, KERNEL_TYPE1, KERNEL_TYPE2, OPT_ALGO,
                   OPT_DEFER, OPT_FUSE, OPT_GRAPH, OPT_FUSED_GRID, OPT_POWER_ITERS, OPT_PROFILE, OPT_WORK_LAYOUT, PlsHipError, lib)
from .model import Group, Handle, Model, as_colmajor, colmajor_empty

SEED_DEFAULT = 0x504C5301  # synthetic-input seed (DESIGN.md "Synthetic inputs")

__all__ = ["Model", "Handle", "Group", "PlsHipError", "lib", "as_colmajor", "colmajor_empty",
           "KERNEL_TYPE1", "KERNEL_TYPE2", "ALGO_KERNEL", "ALGO_NIPALS", "ALGO_GRAM", "ALGO_AUTO", "F64", "F32",
           "OPT_ALGO", "OPT_FUSE", "OPT_PROFILE", "OPT_POWER_ITERS", "OPT_FUSED_GRID", "OPT_WORK_LAYOUT", "OPT_DEFER", "OPT_GRAPH", "SEED_DEFAULT"]
