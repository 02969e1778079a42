"""The stream budget, end to end: the caller on the default stream, a SECOND torch stream alive and used beside it (what
torch.distributed's NCCL backend adds to a process), and the library with three streams (coupled chain on) or two (off).
python tools/probe_stream_budget.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
def factor_ms(n):
    d = 8
    rng = np.random.default_rng(1234)
    x = torch.from_numpy(rng.random((n, d))).cuda()
    hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
    spec = make_spec([0], [0], [d + 1])
    kl = ops.empty(n, n); invd = ops.potrf_workspace(n, torch.float64); info = torch.zeros(1, dtype=torch.int32, device="cuda")
    def run():
        ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7); ops.potrf(kl, invd, info)
    run(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best, ops.last_coupled_panels()
for n in (8192, 16384):
    print("n=%d  library streams 3 (coupled), no other stream : %.2f ms (coupled panels %d)" % ((n,) + factor_ms(n)), flush=True)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    z = torch.ones(1024, device="cuda") * 2          # the second torch stream now has a hardware queue
torch.cuda.synchronize()
for n in (8192, 16384):
    print("n=%d  library streams 3 (coupled), second torch stream alive: %.2f ms (coupled panels %d)" % ((n,) + factor_ms(n)), flush=True)
side2 = torch.cuda.Stream()
with torch.cuda.stream(side2):
    z2 = torch.ones(1024, device="cuda") * 3
torch.cuda.synchronize()
for n in (8192, 16384):
    print("n=%d  library streams 3 (coupled), TWO more torch streams alive: %.2f ms (coupled panels %d)" % ((n,) + factor_ms(n)), flush=True)
side3 = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(side3):
    z3 = torch.ones(1024, device="cuda") * 3
torch.cuda.synchronize()
for n in (8192, 16384):
    print("n=%d  library streams 3 (coupled), THREE more torch streams alive (one high-priority): %.2f ms (coupled panels %d)" % ((n,) + factor_ms(n)), flush=True)
ops.set_coupled_chain(0)
for n in (8192, 16384):
    print("n=%d  library streams 2 (classic), second torch stream alive: %.2f ms (coupled panels %d)" % ((n,) + factor_ms(n)), flush=True)
ops.set_coupled_chain(1)
print("chain back on:", ops.coupled_chain())
