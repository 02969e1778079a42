"""Lower-tile SYRK C -= P P^T (the Cholesky's trailing update) on the caller's stream or, under PG_RAW_STREAM=upd|bg, on the
handle's CU-masked streams: what the mask alone costs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
from pygpr_amd._lib import GEMM_NT
ops = get_ops()
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
g = torch.Generator(device="cuda").manual_seed(1)
for n in (14336, 8192, 4096):
    c = torch.randn(n, n, device="cuda", dtype=torch.float64, generator=g)
    for K in (512, 1024, 2048):
        p = torch.randn(n, K, device="cuda", dtype=torch.float64, generator=g)
        t = ev(lambda: ops.gemm_raw(GEMM_NT, n, n, K, -1.0, p, p, 1.0, c, tri=1))
        print(f"stream={os.environ.get('PG_RAW_STREAM', 'caller')} syrk lower n={n} K={K}: {t:.3f} ms {n*(n+128)*K/t/1e9:.1f} TF/s", flush=True)
    del c
