"""Software overhead of the per-component all-reduce paths at world size 1 (no network time): C3/8 shape
(N = 131072, K = 512, A = 20, NIPALS fused) with no reducer, the torch.distributed (nccl) reducer and the
library-owned RCCL reducer.  Run on one GPU:  python tools/reducer_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist, pls_amd
from pls_amd.distributed import attach_reducer, attach_rccl_reducer, detach_rccl_reducer
N, K, M, A = 131072, 512, 1, 20
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
def run(h, label):
    h.set_option(pls_amd.OPT_ALGO, pls_amd.ALGO_NIPALS)
    X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
    out = h.fit_device(X, Y, A)
    for _ in range(3): h.fit_device(X, Y, A, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): h.fit_device(X, Y, A, out=out)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{label:28s}: {1e3*(t2-t0)/20:.3f} ms per fit (host enqueue {1e3*(t1-t0)/20:.3f} ms), {A*20/(t2-t0):.0f} components/s", flush=True)
h = pls_amd.Handle(); run(h, "no reducer")
h = pls_amd.Handle(); attach_reducer(h, K, M); run(h, "torch.distributed all_reduce")
h = pls_amd.Handle(); comm = attach_rccl_reducer(h); run(h, "library RCCL reducer")
dist.destroy_process_group()
