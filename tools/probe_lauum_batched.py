"""K^-1 = L^-T L^-1 for nexp experts of n points: the batched call (pg_lauum_batched: grid.y = expert) against the same matrices one after
the other (pg_lauum), by HIP events.   python tools/probe_lauum_batched.py [nexp n ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
ops = get_ops()
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
args = [int(a) for a in sys.argv[1:]] or [8, 4096, 8, 2048, 4, 4096, 2, 8192]
for nexp, n in zip(args[0::2], args[1::2]):
    g = torch.Generator(device="cuda").manual_seed(1)
    m = torch.tril(torch.randn(nexp, n, n, device="cuda", dtype=torch.float64, generator=g)) * 1e-2
    k = torch.zeros(nexp, n, n, device="cuda", dtype=torch.float64)
    tb = ev(lambda: ops.lauum_batched(m, k))
    ts = ev(lambda: [ops.lauum(m[e], k[e]) for e in range(nexp)])
    fl = nexp * n ** 3 / 3.0
    print(f"{nexp} x {n}: batched {tb:.3f} ms ({fl/tb/1e9:.1f} TFLOP/s)   one after the other {ts:.3f} ms ({fl/ts/1e9:.1f} TFLOP/s)", flush=True)
    del m, k
