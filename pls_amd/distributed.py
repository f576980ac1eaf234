"""Row (sample) sharding of a fit across the GPUs of one node: SURVEY.md section 8(e).

One process per GPU.  Rank r owns a contiguous block of rows of X and Y; every O(N*K) product
is a sum over rows, so the only exchange is an all-reduce of the K x M partial of X^T Y (once)
and of the packed (K+1)-vector [X^T t, t^T t] (once per component), each as 8 fixed-order slices.
Those messages are a few tens of KB:
latency-bound, so they go through torch.distributed's all_reduce (RCCL over xGMI with the "nccl"
backend) on a buffer this module allocates and hands to the library, stream-ordered with the
library's kernels -- no host synchronisation inside the A-loop on the nccl path.
"""
from __future__ import annotations

import ctypes

from . import _lib as L


def row_partition(n_rows: int, world_size: int, rank: int) -> tuple[int, int]:
    """(row0, nrows) of rank's contiguous block; the first n_rows % world_size ranks get one
    extra row.  Blocks are disjoint, ordered by rank and cover [0, n_rows)."""
    if world_size < 1 or not (0 <= rank < world_size) or n_rows < 0:
        raise ValueError("bad partition arguments")
    base, extra = divmod(n_rows, world_size)
    row0 = rank * base + min(rank, extra)
    return row0, base + (1 if rank < extra else 0)


def attach_reducer(handle, K: int, M: int, group=None, post=None):
    """Give `handle` an all-reduce over torch.distributed's default (or the given) group.

    nccl backend: dist.all_reduce on the CUDA staging tensor, enqueued behind the library's
    kernels on the current stream.  gloo backend (CPU rehearsal of N>1, or several ranks sharing
    one GPU in tests): the staging tensor is bounced through host memory.
    post(view, call_index): testing aid, called on the reduced device view after every collective (fault injection).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    count = L.REDUCE_SLICES * max(K * M, K + 1)
    dev = torch.device("cuda", handle.device)
    stage = torch.zeros(count, dtype=torch.float64, device=dev)
    backend = dist.get_backend(group)
    host = {"buf": torch.zeros(count, dtype=torch.float64).pin_memory()} if backend != "nccl" else None
    base = stage.data_ptr()

    class _Raw:  # zero-copy view of a library-owned device buffer (CUDA array interface, v2)
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    streams = {}  # hipStream_t -> torch view of it (the library's launch stream, whatever torch's current one is)
    calls = [0]

    def _cb(_user, buf, n, _stream):
        try:
            off = (int(buf) - base) // 8
            if 0 <= off and off + n <= count and (int(buf) - base) % 8 == 0:
                view = stage[off:off + n]           # the staging tensor handed over with set_reduce_buffer
            else:                                   # other reductions (X^T X blocks, column statistics, SSE)
                view = torch.as_tensor(_Raw(int(buf), int(n)), device=dev)
            # The contract (include/pls_hip.h): the reduction is ordered on the stream the library passes -- the
            # handle's launch stream, which need not be torch's current stream at callback time.
            key = int(_stream or 0)
            st = streams.get(key)
            if st is None:
                st = streams[key] = torch.cuda.ExternalStream(key, device=dev) if key else torch.cuda.default_stream(dev)
            with torch.cuda.stream(st):
                if backend == "nccl":
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
                else:
                    if host["buf"].numel() < n:  # bounce buffer grows with the largest message
                        host["buf"] = torch.zeros(int(n), dtype=torch.float64).pin_memory()
                    hb = host["buf"][:n]
                    st.synchronize()
                    hb.copy_(view)
                    dist.all_reduce(hb, op=dist.ReduceOp.SUM, group=group)
                    view.copy_(hb, non_blocking=False)
                    st.synchronize()
                if post is not None:
                    post(view, calls[0])
                calls[0] += 1
            return 0
        except Exception:  # an exception must not unwind through the C frame
            import traceback
            traceback.print_exc()
            return 1

    cb = L.ALLREDUCE_FN(_cb)
    handle._keep += [cb, stage, host]
    L.check(handle._lib.pls_hip_set_reduce_buffer(handle.h, ctypes.c_void_p(base), count), handle.h)
    L.check(handle._lib.pls_hip_set_reducer(handle.h, cb, None, rank, world), handle.h)
    return stage


def make_host_allreduce(group=None):
    """all-reduce of a float64 numpy view through torch.distributed (gloo): the same reduction the
    device path performs, for CPU rehearsals of the sharded algorithm."""
    import torch
    import torch.distributed as dist

    def _allreduce(view):
        t = torch.from_numpy(view)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    return _allreduce


def attach_rccl_reducer(handle, group=None):
    """Alternative to attach_reducer: the library's own RCCL communicator (libpls_hip_rccl.so,
    include/pls_hip_rccl.h) -- ncclAllReduce is issued directly on the handle's stream, no Python in the
    A-loop.  torch.distributed is used once, to hand the ncclUniqueId of rank 0 to the other ranks.
    Returns the communicator (keep it; detach_rccl_reducer releases it)."""
    import os

    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    # Phase 1 (local, may fail on one rank only): load the helper library, rank 0 draws the unique id.  The ranks agree
    # on the outcome BEFORE the first collective that depends on it, so a local failure raises on every rank instead of
    # leaving the others in a broadcast nobody serves.
    lib, ident, err = None, ctypes.create_string_buffer(128), None
    try:
        lib = ctypes.CDLL(os.path.join(os.path.dirname(L.LIB_PATH), "libpls_hip_rccl.so"), mode=ctypes.RTLD_GLOBAL)
        lib.pls_hip_rccl_unique_id.argtypes = [ctypes.c_void_p]
        lib.pls_hip_rccl_attach.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_void_p)]
        lib.pls_hip_rccl_detach.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        if rank == 0:
            L.check(lib.pls_hip_rccl_unique_id(ident))
    except Exception as e:  # noqa: BLE001
        err = e
    if world > 1:
        import torch
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            raise RuntimeError("RCCL helper unavailable on " + ("this rank: " + repr(err) if err else "another rank"))
        box = [bytes(ident.raw)]
        dist.broadcast_object_list(box, src=0, group=group)
        ident = ctypes.create_string_buffer(box[0], 128)
    elif err:
        raise err
    comm = ctypes.c_void_p()
    L.check(lib.pls_hip_rccl_attach(handle.h, handle.device, ident, rank, world, ctypes.byref(comm)), handle.h)
    handle._keep += [lib]
    handle._rccl = (lib, comm)
    return comm


def rccl_comm_count(handle) -> int:
    """ranks of the attached RCCL communicator as RCCL reports them (ncclCommCount)"""
    lib, comm = handle._rccl
    lib.pls_hip_rccl_comm_count.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    n = ctypes.c_int(0)
    L.check(lib.pls_hip_rccl_comm_count(comm, ctypes.byref(n)), handle.h)
    return int(n.value)


def detach_rccl_reducer(handle):
    lib, comm = handle._rccl
    L.check(lib.pls_hip_rccl_detach(handle.h, comm), handle.h)
    handle._rccl = None


def attach_ipc_exchange(handle, group=None):
    """The library's own device-side exchange between the ranks' GPUs (include/pls_hip.h, pls_hip_xchg_*): every rank
    writes its partial sums straight into the other ranks' inboxes (opened over IPC) and spins on sequence flags -- two
    small launches per rank and collective, no RCCL, nothing on the host.  torch.distributed is used three times, at
    set-up: to all-gather the 160-byte IPC handles and to let the ranks agree that every step worked.  Ends with a
    collective self-test; raises on EVERY rank (after releasing the exchange) if any rank could not set it up, so that
    the caller can fall back to attach_rccl_reducer / attach_reducer on all of them."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lib = handle._lib
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"

    def agree(ok: bool, what: str, why: str = ""):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            lib.pls_hip_xchg_destroy(handle.h)
            raise RuntimeError(f"device-side exchange unavailable ({what}" + (f": {why}" if why else " on another rank") + ")")

    mine = ctypes.create_string_buffer(L.XCHG_HANDLE_BYTES)
    rc = lib.pls_hip_xchg_create(handle.h, rank, world, mine)
    agree(rc == L.OK, "create", "" if rc == L.OK else L.last_error(handle.h))
    blobs = [None] * world
    dist.all_gather_object(blobs, bytes(mine.raw), group=group)
    allb = ctypes.create_string_buffer(b"".join(blobs), L.XCHG_HANDLE_BYTES * world)
    rc = lib.pls_hip_xchg_connect(handle.h, allb)
    agree(rc == L.OK, "connect", "" if rc == L.OK else L.last_error(handle.h))
    rc = lib.pls_hip_xchg_selftest(handle.h)
    agree(rc == L.OK, "self-test", "" if rc == L.OK else L.last_error(handle.h))


def detach_ipc_exchange(handle):
    L.check(handle._lib.pls_hip_xchg_destroy(handle.h), handle.h)
