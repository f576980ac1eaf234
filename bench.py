#!/usr/bin/env python3
"""bench.py -- the PLS fit hot path on MI355X: NIPALS components/sec + achieved HBM GB/s.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete fit (A components) of BASELINE.json config 3 -- synthetic tall
n = 1,048,576 x p = 512, m = 1, A = 20, fp64 -- with X, Y resident in HBM before the timed
region.  With N > 1 the SAME matrix is row-sharded over the ranks (strong scaling: the metric
is quoted at n = 1M for 1/2/4/8 GPUs), one process per GPU, the K x M / (K+1)-length partial
products all-reduced with torch.distributed (RCCL over xGMI).  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel family of the timed fits: algorithmic bytes per launch / average
                launch duration (HIP events on the launch stream, recorded by the library while
                the timed steps run) against the 8.0 TB/s HBM3E peak.
  cpu_baseline  the oracle's C restatement of the reference algorithm (Eigen is unavailable,
                so this is "kind": "port"), one host core, on a bounded row sample.
  alt           the other execution plans on the same data (not the headline value).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md:36
WORKLOADS = {
    # name: (N, K, M, A, dtype)
    "C3": (1 << 20, 512, 1, 20, "f64"),
    "C4": (131072, 4096, 8, 50, "f32"),
    "C5rank": (2097152, 1024, 4, 20, "f64"),  # one rank's shard of config 5
    "C3eighth": (131072, 512, 1, 20, "f64"),  # one rank's share of config 3 on 8 GPUs (overhead study)
    "tiny": (4096, 64, 1, 5, "f64"),
    # shapes that used to fall off the one-sweep kernels (ld = N exactly: odd N leaves every second column at 8 mod 16)
    "C3odd-": (1048575, 512, 1, 20, "f64"),
    "C3odd+": (1048577, 512, 1, 20, "f64"),
    "tall64": (1 << 24, 64, 1, 20, "f64"),   # 32 column groups of one matrix span 4.3 GB: per-wave descriptors
    "tall64a": (8388544, 64, 1, 20, "f64"),  # the tallest 64-column matrix one descriptor per column-group set still covers
    "C4odd": (131071, 4096, 8, 50, "f32"),
    "C4m16": (131072, 4096, 16, 50, "f32"), "C4m32": (131072, 4096, 32, 50, "f32"),  # more responses than the cooperative update takes
    # narrow matrices, 4.3 GB each (few columns per lane in the resident tile)
    "narrow32": (1 << 24, 32, 1, 20, "f64"), "narrow128": (1 << 22, 128, 1, 20, "f64"), "narrow256": (1 << 21, 256, 1, 20, "f64"),
}
WORKLOADS.update({
    # beyond 4096 columns: row-pack tiles (512 column groups), 2.1 / 4.3 GB
    # few rows, very many columns (beyond every resident tile): the one-product kernels
    "shortwide": (4096, 32768, 1, 10, "f64"), "genes": (512, 50000, 1, 10, "f64"), "genes4": (512, 50000, 4, 10, "f64"),
    "wide16k": (32768, 16384, 8, 30, "f32"), "wide12k64": (43690, 12288, 1, 20, "f64"),  # (KERNEL plan: 32 columns per lane)
    "C3m4": (1 << 20, 512, 4, 20, "f64"),  # config 3 with four responses (SYRK with X^T Y of several responses on board)
    "mid1": (1000, 4000, 1, 10, "f64"), "mid2": (2000, 2000, 1, 10, "f64"), "mid3": (500, 10000, 1, 10, "f64"),  # 32-40 MB: launch-bound
    "mid4": (500, 6000, 1, 10, "f64"), "mid5": (1500, 10000, 1, 10, "f64"), "mid6": (3000, 12000, 1, 10, "f64"), "mid7": (6000, 10000, 1, 10, "f64"),
    "abc": (100000, 40, 12, 40, "f64"),  # few predictors, a dozen responses, every component (the reference author's own use)
    "wide8k1": (65536, 8192, 1, 20, "f32"), "wide8k2": (65536, 8192, 2, 20, "f32"),
    "wide8k": (65536, 8192, 8, 30, "f32"), "wide6k64": (87381, 6144, 1, 20, "f64"), "wide8k64": (65536, 8192, 2, 20, "f64"),
})
WORKLOADS.update({
    # config 3 with 8 components (fewer than the copy into tiles pays for: the KERNEL plan reads the caller's matrix every
    # component), and the same with the columns NOT a power of two apart (ld = N + 64: 8 MiB + 512 bytes)
    "C3a8": (1 << 20, 512, 1, 8, "f64"), "C3a8pad": (1 << 20, 512, 1, 8, "f64"), "C3pad": (1 << 20, 512, 1, 20, "f64"),
})
TIGHT_LD = {"C3odd-", "C3odd+", "C4odd"}  # leading dimension = N (no padding to 16 bytes)
PAD_LD = {"C3a8pad": 64, "C3pad": 64}      # leading dimension = N + this many elements


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--algo", default="nipals", choices=["nipals", "kernel", "gram"],
                    help="nipals: north-star sequence with the rank-1 deflation of X (headline); "
                         "kernel: the reference's own sequence, X read-only")
    ap.add_argument("--fuse", type=int, default=1)
    ap.add_argument("--profile-after", action="store_true",
                    help="N = 1: time the steps WITHOUT the HIP-event brackets around every streaming launch (as every N > 1 "
                         "run does) and take the roofline from extra profiled steps afterwards; default: brackets in the "
                         "timed region itself")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (gloo: rehearsal with ranks sharing one GPU)")
    ap.add_argument("--reducer", default=None, choices=["torch", "rccl", "ipc"],
                    help="N > 1: ipc (default with --backend nccl) = the library's device-side exchange, every rank writing "
                         "its partial sums straight into the other ranks' inboxes over xGMI (pls_hip_xchg_*, no RCCL in the "
                         "component loop); falls back -- on every rank together -- to rccl = the library's own RCCL "
                         "communicator, ncclAllReduce issued on the launch stream (include/pls_hip_rccl.h), and from there "
                         "to torch (default with --backend gloo) = torch.distributed all_reduce from a ctypes callback")
    ap.add_argument("--no-alt", action="store_true", help="skip the alternative execution plans")
    ap.add_argument("--defer", type=int, default=1, choices=[1, 2, 3, 4],
                    help="NIPALS plan: write the deflated matrix back every D-th component only (default 1 = explicit "
                         "deflation every component, the headline)")
    ap.add_argument("--rccl-leg-at-one-rank", action="store_true",
                    help="rehearsal: run the RCCL leg (alt.rccl) with ONE rank as well -- a 1-rank nccl process group and "
                         "communicator -- so that its code path can be exercised on a single GPU")
    ap.add_argument("--alt-timeout", type=float, default=90.0, help="N > 1: seconds the RCCL leg (alt.rccl) may take before the "
                    "line is printed without it")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-rows", type=int, default=1 << 20, help="rows of the CPU baseline sample (default: the whole workload, ~10 s on one core)")
    return ap.parse_args()


def timed_fits(h, torch, dist, world, X, Y, A, steps, warmup, out):
    for _ in range(warmup):
        h.fit_device(X, Y, A, out=out)
    torch.cuda.synchronize()
    h.timing()  # drop warm-up events
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.fit_device(X, Y, A, out=out)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el, h.timing()


def pmc_traffic(workload, algo, fuse, family):
    """HBM bytes per launch of `family` from the committed rocprofv3 PMC summary of this configuration
    (profiles/rNN/pmc_traffic_<workload>_<algo>_<fused|unfused>.json, written by
    tools/summarize_pmc.py from separate FETCH_SIZE / WRITE_SIZE passes); None if there is none."""
    import glob
    name = f"pmc_traffic_{workload}_{algo}_{'fused' if fuse else 'unfused'}.json"
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    if not cands:
        return None, None
    d = json.load(open(cands[-1]))
    pref = {"fused": "fused_pass_kernel", "deflate": "deflate", "xb": "xb_kernel", "xty": "xty_kernel"}[family]
    rows = [v for k, v in d.items() if k.startswith(pref)]
    if not rows:
        return None, None
    n = sum(r["launches_fetch_pass"] for r in rows)
    return int(sum(r["hbm_bytes_per_launch"] * r["launches_fetch_pass"] for r in rows) / n), os.path.relpath(cands[-1], ROOT)


def roofline_of(tm):
    fams = [f for f in ("fused", "deflate", "xb", "xty") if tm["launches"][f] > 0]
    if not fams:
        return None
    dom = max(fams, key=lambda f: tm["ms"][f])
    n = tm["launches"][dom]
    avg_ms = tm["ms"][dom] / n
    bytes_per = tm["bytes"][dom] / n
    ach = bytes_per / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
            "frac_of_measured_copy_ceiling": round(ach / 6290.0, 4),  # 6.29 TB/s float4 copy, MI355X_MICROARCH.md:36
            "launches": n, "avg_launch_ms": round(avg_ms, 5), "algorithmic_bytes_per_launch": int(bytes_per),
            "families_ms_per_fit": {f: round(tm["ms"][f] / max(tm["fits"], 1), 4) for f in tm["ms"]}}


FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X datasheet, fp64 matrix cores (the guide has no fp64 figure)


def syrk_roofline(N, K, syrk_ms):
    """The X^T X launch of the GRAM / KERNEL_TYPE2 plans on the MFMA roofline: flops the kernel EXECUTES (16 x 16 tiles:
    the 128 x 128 blocks above the diagonal in full, of every diagonal block the 36 of 64 tiles on or above its diagonal
    ) / launch time."""
    if not syrk_ms or syrk_ms <= 0:
        return None
    nbk = (K + 127) // 128
    diag_tiles = 36
    tiles = 64 * (nbk * (nbk - 1) // 2) + diag_tiles * nbk
    executed = 2.0 * N * 16 * 16 * tiles
    ach = executed / (syrk_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "syrk (X^T X, fp64 MFMA 16x16x4)", "achieved": round(ach, 2), "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(ach / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": None, "avg_launch_ms": round(syrk_ms, 4),
            "executed_gflop_per_launch": round(executed / 1e9, 2),
            "nominal_tflops_2NK2_symmetry_counted": round(2.0 * N * K * K / (syrk_ms * 1e-3) / 1e12, 2)}


def cpu_share():
    """host cores this process may really use: the affinity mask, cut by a cgroup CPU quota (the GPU boxes of this pool give a
    job the share of one GPU -- 16 of 256 hardware threads -- which omp_get_max_threads does not see)"""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(N, K, M, A, rows):
    """oracle C restatement (reference operation sequence: 1 + 2A passes over X), 1 core, -O3."""
    from oracle import pls_oracle as po
    rows = min(rows, N)
    os.environ.setdefault("OMP_PROC_BIND", "close")  # (threads stay where they first touched their row blocks)
    os.environ.setdefault("OMP_PLACES", "cores")
    gen = po.OracleLib(omp=True)
    share = cpu_share()
    gen.set_num_threads(share)
    Xh = gen.synth_x(0, rows, K)
    Yh = gen.synth_y(0, rows, M)
    one = po.OracleLib(omp=False)
    t0 = time.perf_counter()
    one.plsr(Xh, Yh, A)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    gen.plsr(Xh, Yh, A)
    tn = time.perf_counter() - t0
    scale = rows / N  # cost is linear in N: components/s at the full N = sample rate x rows/N
    base = {"value": round(A / t1 * scale, 4), "unit": "components/s", "cores": 1, "kind": "port",
            "sample": f"first {rows} of {N} rows x {K} cols, A={A}, fp64, one fit = {t1:.2f} s; "
                      f"rate scaled by {rows}/{N} (cost linear in rows); restatement of the reference "
                      f"algorithm, Eigen unavailable"}
    import glob
    nodes = len(glob.glob("/sys/devices/system/node/node[0-9]*")) or 1
    omp = {"value": round(A / tn * scale, 4), "unit": "components/s", "cores": gen.num_threads(), "kind": "port",
           "sample": f"same sample, -O3 -march=x86-64-v3 -fopenmp, {gen.num_threads()} OpenMP threads (the CPU share of this job: "
                     f"affinity {len(os.sched_getaffinity(0))}, cgroup quota -> {share}) on a host with {nodes} NUMA node(s), "
                     f"every sweep of X partitioned by row blocks and X generated under the same partition (first-touch: the "
                     f"pages of a block live where their reader runs), one fit = {tn:.2f} s"}
    return base, omp


def rccl_leg(h, a, torch, dist, world, X, Y, A, out, reducer_used):
    """N > 1: a short second leg of the same sharded fit under the library's RCCL reducer (ncclAllReduce issued on the
    launch stream between the kernels -- the north star's "RCCL all-reduce over xGMI"), whatever reducer the headline ran
    with, so that every scaling run carries an RCCL figure and the rank count RCCL itself reports.  Every rank takes
    the same branches (the outcome of each step is agreed through torch.distributed).  Returns the dict for alt.rccl."""
    from pls_amd.distributed import attach_rccl_reducer, detach_ipc_exchange, detach_rccl_reducer, rccl_comm_count
    if a.backend != "nccl":
        return {"unavailable": "rehearsal backend gloo: the ranks share one GPU, which RCCL refuses (one rank per device)"}
    import pls_amd
    already = reducer_used.startswith("rccl")
    try:
        if reducer_used.startswith("ipc"):
            detach_ipc_exchange(h)
        if not already:
            if reducer_used.startswith("torch"):
                h.clear_reducer()
            attach_rccl_reducer(h)  # (with one rank: a 1-rank communicator)
        ok, why = 1, ""
    except Exception as e:  # noqa: BLE001
        ok, why = 0, repr(e)
    flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if ok and not already:
            detach_rccl_reducer(h)
        return {"unavailable": "library RCCL communicator could not be set up" + (": " + why if why else " on another rank")}
    nranks = rccl_comm_count(h)
    h.set_option(pls_amd.OPT_PROFILE, 0)
    steps = min(a.steps, 3)
    el, _ = timed_fits(h, torch, dist, world, X, Y, A, steps, 1, out)
    h.synchronize()
    return {"components_per_s": round(A * steps / el, 2), "ms_per_fit": round(el / steps * 1e3, 3), "steps": steps,
            "nranks_reported_by_rccl": nranks, "reducer": "ncclAllReduce(ncclDouble, ncclSum) on the launch stream, "
            "8*K*M values once + 8*(K+1) values per component (include/pls_hip_rccl.h)"}


def spawn_ranks(a):
    """`python bench.py --gpus N` outside a launcher: start the N ranks the way the driver does (one process per
    GPU through torch.distributed.run) as a CHILD process -- nothing in this process has touched the GPU yet --
    and exit with its status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    if a.reducer is None:
        a.reducer = "ipc" if a.backend == "nccl" else "torch"
    import torch
    import torch.distributed as dist

    import pls_amd
    from pls_amd.distributed import attach_reducer, row_partition

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    local = local if a.backend == "nccl" else local % max(ndev, 1)
    torch.cuda.set_device(local)
    if world > 1 or a.rccl_leg_at_one_rank:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:  # a free port: two bench / test processes on one box must not collide
                import socket

                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    N, K, M, A, dt = WORKLOADS[a.workload]
    tdt = torch.float64 if dt == "f64" else torch.float32
    row0, nrows = row_partition(N, world, rank)
    h = pls_amd.Handle()
    X = h.synth_x(row0, nrows, K, pls_amd.SEED_DEFAULT, dtype=tdt)
    Y = h.synth_y(row0, nrows, M, pls_amd.SEED_DEFAULT, dtype=tdt)
    if a.workload in TIGHT_LD:  # the reference's own layout: Eigen matrices have ld = rows
        Xt = pls_amd.colmajor_empty(nrows, K, tdt, X.device, ld=nrows); Xt.copy_(X); X = Xt
        Yt = pls_amd.colmajor_empty(nrows, M, tdt, Y.device, ld=nrows); Yt.copy_(Y); Y = Yt
    if a.workload in PAD_LD:
        Xt = pls_amd.colmajor_empty(nrows, K, tdt, X.device, ld=nrows + PAD_LD[a.workload]); Xt.copy_(X); X = Xt
    reducer_used = None
    if world > 1:
        reducer_used = a.reducer
        notes = []
        if reducer_used == "ipc":
            # the device-side exchange sets itself up collectively (self-test included) and raises on EVERY rank if any
            # rank cannot do it: then all of them move on to RCCL -- a scaling run must not die here
            from pls_amd.distributed import attach_ipc_exchange
            try:
                attach_ipc_exchange(h)
            except Exception as e:  # noqa: BLE001
                notes.append("device-side exchange unavailable: " + repr(e))
                reducer_used = "rccl" if a.backend == "nccl" else "torch"
        if reducer_used == "rccl":
            # the library's own communicator; if it cannot be set up on EVERY rank (the ranks agree on that through
            # torch.distributed), all of them fall back to the torch reducer -- a scaling run must not die here
            from pls_amd.distributed import attach_rccl_reducer
            why = ""
            try:
                attach_rccl_reducer(h)
                ok = 1
            except Exception as e:  # noqa: BLE001
                ok, why = 0, repr(e)
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if ok:
                    from pls_amd.distributed import detach_rccl_reducer
                    detach_rccl_reducer(h)
                attach_reducer(h, K, M)
                reducer_used = "torch (library RCCL communicator unavailable" + (": " + why if why else " on another rank") + ")"
        elif reducer_used == "torch":
            attach_reducer(h, K, M)
        if notes:
            reducer_used += " (" + "; ".join(notes) + ")"
    algo = {"nipals": pls_amd.ALGO_NIPALS, "kernel": pls_amd.ALGO_KERNEL, "gram": pls_amd.ALGO_GRAM}[a.algo]
    h.set_option(pls_amd.OPT_ALGO, algo)
    h.set_option(pls_amd.OPT_FUSE, a.fuse)
    h.set_option(pls_amd.OPT_DEFER, a.defer)
    # HIP events around every streaming launch (the roofline figures).  At N = 1 they bracket the launches of the
    # timed steps themselves; the brackets cost ~8 us per component (0.5 % of a single-GPU component, but 4 % of a
    # 1/8-size one), so at N > 1 the timed steps run without them and the roofline comes from extra profiled steps
    # after the timed region.
    after = world > 1 or a.profile_after
    h.set_option(pls_amd.OPT_PROFILE, 0 if after else 1)
    out = None
    if a.workload in TIGHT_LD:  # the scores in the reference's layout as well (T is N x A with ld = N)
        f64 = torch.float64
        out = {k: pls_amd.colmajor_empty(K, A, f64, X.device, ld=K) for k in "WPR"}
        out["Q"] = pls_amd.colmajor_empty(M, A, f64, X.device, ld=M)
        out["B"] = pls_amd.colmajor_empty(K, M, f64, X.device, ld=K)
        out["T"] = pls_amd.colmajor_empty(nrows, A, tdt, X.device, ld=nrows)
    def measure(out):
        out = h.fit_device(X, Y, A, out=out)  # allocates outputs + workspace once
        torch.cuda.synchronize()
        h.set_option(pls_amd.OPT_PROFILE, 0 if after else 1)
        el, tm = timed_fits(h, torch, dist, world, X, Y, A, a.steps, a.warmup, out)
        where = "timed steps"
        if after:
            h.set_option(pls_amd.OPT_PROFILE, 1)
            _, tm = timed_fits(h, torch, dist, world, X, Y, A, min(a.steps, 3), 1, out)
            where = "separate profiled steps after the timed region (N > 1, or --profile-after)"
        h.synchronize()  # (raises if the ranks of a sharded fit diverged or the exchange timed out: include/pls_hip.h)
        return out, el, tm, where

    if world == 1:
        out, el, tm, roofline_where = measure(out)
    else:
        # A reducer that set itself up (self-test included) can still fail in the fits -- the device-side exchange has never
        # crossed a real xGMI link in this pipeline.  A scaling run must not die of that: the ranks agree on the outcome and,
        # if any of them failed, ALL drop the reducer they had, take the torch reducer and measure again; the line says so.
        why = ""
        try:
            res = measure(out)
            ok = 1
        except Exception as e:  # noqa: BLE001
            ok, why, res = 0, repr(e), None
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            had = str(reducer_used)
            try:
                if had.startswith("ipc"):
                    from pls_amd.distributed import detach_ipc_exchange
                    detach_ipc_exchange(h)
                elif had.startswith("rccl"):
                    from pls_amd.distributed import detach_rccl_reducer
                    detach_rccl_reducer(h)
            except Exception as e:  # noqa: BLE001
                why += "; detach: " + repr(e)
            if had.startswith("torch"):
                raise SystemExit("the torch reducer failed in the fits: " + (why or "on another rank"))
            for _ in range(3):  # flush what the failed fits left behind (the time-out's status, then the replica guard's)
                try:
                    h.synchronize()
                    break
                except Exception:  # noqa: BLE001
                    pass
            attach_reducer(h, K, M)
            reducer_used = "torch (the " + had + " reducer failed in the fits" + (": " + why[:300] if why else " on another rank") + ")"
            res = measure(None)
        out, el, tm, roofline_where = res
    value = A * a.steps / el
    es = 8 if dt == "f64" else 4
    line = {
        "metric": "NIPALS components/sec", "value": round(value, 2), "unit": "components/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(el / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dt, "data": "synthetic",
        "config": {"workload": f"{a.workload}: synthetic tall n={N} x p={K}, m={M}, A={A}, {dt}, resident in HBM",
                   "algo": a.algo, "fuse": a.fuse, "deflation_written_back_every": a.defer, "rows_per_gpu": nrows,
                   "parallelism": f"row-shard x{world}" if world > 1 else "single GPU",
                   "reducer": reducer_used},
        "roofline": roofline_of(tm),
        # end-to-end effective stream rate: (2A) N K s / t_fit, the fused lower bound
        "effective_gbs": round(2 * A * N * K * es / (el / a.steps) / 1e9, 1),
    }

    if line["roofline"] is not None:
        line["roofline"]["measured_over"] = roofline_where
        if world == 1:  # (the contract: the launch durations come from the timed steps themselves)
            line["roofline"]["note"] = ("HIP-event brackets around every streaming launch sit INSIDE the timed region at N = 1 "
                                        "(about 8 us per component): `value` is ~0.5 % below the un-bracketed rate")
    if line["roofline"] is not None and world == 1:  # the PMC summary was taken on the single-GPU shape
        tr, src = pmc_traffic(a.workload, a.algo, a.fuse, line["roofline"]["kernel"])
        line["roofline"]["traffic"] = tr
        line["roofline"]["traffic_source"] = src
    if rank == 0 or world > 1:
        alt = {}
        if not a.no_alt and world == 1:
            plans = {"kernel_fused": (0, 1), "kernel_unfused": (0, 0), "nipals_fused": (1, 1), "nipals_unfused": (1, 0)}
            if K <= 2048:
                plans["gram_mfma_syrk"] = (2, 1)  # XX on the matrix cores + component loop on K x K + T = X R
            for name, (al, fu) in plans.items():
                if (al, fu) == (algo, a.fuse):
                    continue
                h.set_option(pls_amd.OPT_ALGO, al)
                h.set_option(pls_amd.OPT_FUSE, fu)
                e2, t2 = timed_fits(h, torch, dist, 1, X, Y, A, max(2, a.steps // 2), 1, out)
                st = max(2, a.steps // 2)
                rl = roofline_of(t2)
                if al == 2 and t2["launches"]["xty"] > 0:  # GRAM: the dominant launch is the matrix-core SYRK, not a stream
                    rl = syrk_roofline(nrows, K, t2["ms"]["xty"] / t2["launches"]["xty"])
                alt[name] = {"components_per_s": round(A * st / e2, 2), "ms_per_fit": round(e2 / st * 1e3, 3), "roofline": rl}
            # opt-in variant of the NIPALS plan: the deflated matrix is written back every D-th component only, the
            # pending rank-1 updates are re-applied in registers (same roundings as the explicit plan; K <= 512)
            if K <= 512:
                h.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_NIPALS)
                h.set_option(pls_amd.OPT_FUSE, 1)
                for D in (2, 4):
                    h.set_option(pls_amd.OPT_DEFER, D)
                    st = max(2, a.steps // 2)
                    e2, t2 = timed_fits(h, torch, dist, 1, X, Y, A, st, 1, out)
                    alt[f"nipals_deferred_writeback_every_{D}"] = {
                        "components_per_s": round(A * st / e2, 2), "ms_per_fit": round(e2 / st * 1e3, 3),
                        "roofline": roofline_of(t2),
                        "note": "X_a materialised in HBM every D-th component; not the headline plan"}
                h.set_option(pls_amd.OPT_DEFER, 1)
            # METHOD::KERNEL_TYPE2 (XX = X^T X on the matrix cores, then no pass over X; T not computed)
            if K <= 2048:
                h.set_option(pls_amd.OPT_ALGO, 0)
                st = max(2, a.steps // 2)
                for _ in range(2):
                    h.fit_device(X, Y, A, method=pls_amd.KERNEL_TYPE2, out=out)
                torch.cuda.synchronize(); h.timing()
                t0 = time.perf_counter()
                for _ in range(st):
                    h.fit_device(X, Y, A, method=pls_amd.KERNEL_TYPE2, out=out)
                torch.cuda.synchronize()
                e2 = time.perf_counter() - t0
                t2 = h.timing()
                n2 = max(t2["launches"]["xty"], 1)
                # the xty family of a KERNEL_TYPE2 fit: the SYRK launch alone when X^T Y rides along in its diagonal
                # workgroups (one launch per fit), otherwise SYRK + the separate X^T Y pass the main plan also has
                syrk_ms = t2["ms"]["xty"] / t2["fits"]
                xy_on_board = t2["launches"]["xty"] <= t2["fits"]
                if not xy_on_board:
                    syrk_ms -= tm["ms"]["xty"] / max(tm["fits"], 1)
                alt["type2_mfma_syrk"] = {"components_per_s": round(A * st / e2, 2), "ms_per_fit": round(e2 / st * 1e3, 3),
                                          "syrk_launch_also_forms_xty": bool(xy_on_board),
                                          "roofline": syrk_roofline(nrows, K, syrk_ms),
                                          "note": "T (scores) not computed by this method; 'executed' counts the 16 x 16 tiles the kernel "
                                                  "computes, 'nominal' the full 2 N K^2 (can exceed the MFMA peak)"}
            # the stand-alone rank-1 deflation (the north star's "deflation step"), X -= t p^T in place
            h.set_option(pls_amd.OPT_ALGO, algo)
            h.set_option(pls_amd.OPT_FUSE, a.fuse)
            t = out["T"][:, 0].contiguous()
            p = out["P"][:, 0].contiguous()
            W = pls_amd.colmajor_empty(nrows, K, tdt, X.device)
            W.copy_(X)
            for _ in range(2):
                h.deflate(W, t, p, dst=W)
            torch.cuda.synchronize()
            h.timing()
            for _ in range(10):
                h.deflate(W, t, p, dst=W)
            td = h.timing()
            alt["deflate_kernel"] = roofline_of(td)
            del W
        line["alt"] = alt

    if (world > 1 or a.rccl_leg_at_one_rank) and not a.no_alt:
        # The RCCL leg must never cost the run its line: if it does not finish in time (a communicator that hangs in its
        # set-up on one rank) rank 0 prints the headline as measured and every rank leaves.
        import threading

        def give_up():
            if rank == 0:
                line.setdefault("alt", {})["rccl"] = {"unavailable": f"RCCL leg did not finish within {a.alt_timeout} s"}
                print(json.dumps(line), flush=True)
            print(f"bench.py: rank {rank}: the RCCL leg hung (> {a.alt_timeout} s); leaving with the headline only", file=sys.stderr, flush=True)
            os._exit(0)

        finished = threading.Event()
        dog = threading.Timer(a.alt_timeout, lambda: None if finished.is_set() else give_up())
        dog.daemon = True
        dog.start()
        try:
            res = rccl_leg(h, a, torch, dist, world, X, Y, A, out, reducer_used or "none")
        except Exception as e:  # noqa: BLE001  (a rank that fails alone leaves the others to the watchdog)
            res = {"unavailable": "RCCL leg raised " + repr(e)}
        finished.set()
        dog.cancel()
        line.setdefault("alt", {})["rccl"] = res

    if rank == 0:
        if not a.no_cpu and world == 1:  # the CPU leg runs on rank 0 at N = 1 only
            try:
                base, omp = cpu_baseline(N, K, M, A, a.cpu_rows)
                line["cpu_baseline"] = base
                line["cpu_baseline_allcore"] = omp
            except Exception as e:  # the oracle is test infrastructure: report, do not hide
                line["cpu_baseline"] = {"value": None, "unit": "components/s", "cores": 1, "kind": "port",
                                        "sample": f"failed: {e!r}"}
        print(json.dumps(line), flush=True)
    if world > 1 or a.rccl_leg_at_one_rank:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
