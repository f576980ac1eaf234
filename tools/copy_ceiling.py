"""Read+write ceiling of this MI355X as seen by vendor code paths: torch copy_ (elementwise kernel) and
hipMemcpyDtoD on 4.3 GB (the size of config 3's X), plus a read-only reduction, for comparison with the
tile kernels (profiles/r1)."""
import torch, time
n = 1048576 * 512
a = torch.randn(n, dtype=torch.float64, device="cuda:0"); b = torch.empty_like(a)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
ms = t(lambda: b.copy_(a)); print(f"torch copy_ 4.3 GB: {ms:.3f} ms  {2*n*8/ms/1e9:.2f} TB/s (read+write)")
ms = t(lambda: a.add_(1.0)); print(f"torch add_ in place: {ms:.3f} ms  {2*n*8/ms/1e9:.2f} TB/s (read+write)")
ms = t(lambda: torch.add(a, 1.0, out=b)); print(f"torch add out of place: {ms:.3f} ms  {2*n*8/ms/1e9:.2f} TB/s (read+write)")
ms = t(lambda: a.sum()); print(f"torch sum: {ms:.3f} ms  {n*8/ms/1e9:.2f} TB/s (read)")
ms = t(lambda: b.fill_(1.0)); print(f"torch fill_: {ms:.3f} ms  {n*8/ms/1e9:.2f} TB/s (write)")
a32 = a.view(torch.float32); b32 = b.view(torch.float32)
ms = t(lambda: b32.copy_(a32)); print(f"torch copy_ fp32 view: {ms:.3f} ms  {2*n*8/ms/1e9:.2f} TB/s (read+write)")
