"""Can two RCCL ranks share one GPU on this box (so that the nccl branch of the reducer could be tested
at N = 2 on a 1-GPU machine)?  Prints the outcome; exits non-zero quickly if RCCL refuses."""
import os, sys, datetime
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", timeout=datetime.timedelta(seconds=40))
    t = torch.full((8,), float(rank + 1), device="cuda:0", dtype=torch.float64)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce on a shared GPU -> {t[0].item()} (expected {world * (world + 1) / 2})", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: RCCL refused two ranks on one device: {type(e).__name__}: {str(e)[:300]}", flush=True)
    sys.exit(3)
