"""K^-1 = M^T M (M = L^-1, lower): the shipped TN form on M against the NT form on M^T (both operands K-contiguous), and what a mirror costs."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
from pygpr_amd._lib import GEMM_NT, GEMM_TN
ops = get_ops()
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
for n in (8192, 16384):
    g = torch.Generator(device="cuda").manual_seed(1)
    m = torch.tril(torch.randn(n, n, device="cuda", dtype=torch.float64, generator=g)) / n ** 0.5
    mt = m.T.contiguous()
    c1 = torch.zeros(n, n, device="cuda", dtype=torch.float64); c2 = torch.zeros_like(c1)
    t_tn = ev(lambda: ops.gemm_raw(GEMM_TN, n, n, n, 1.0, m, m, 0.0, c1, tri=1, klo=1))
    t_nt = ev(lambda: ops.gemm_raw(GEMM_NT, n, n, n, 1.0, mt, mt, 0.0, c2, tri=1, klo=1))
    err = (torch.tril(c1) - torch.tril(c2)).abs().max().item()
    t_sym = ev(lambda: ops.symmetrize(m, n), 2)
    print(f"n={n}: lauum TN on M {t_tn:.3f} ms ({n**3/3/t_tn/1e9:.1f} TF/s) | NT on M^T {t_nt:.3f} ms ({n**3/3/t_nt/1e9:.1f} TF/s) | max diff {err:.2e} | naive mirror {t_sym:.3f} ms", flush=True)
    del m, mt, c1, c2
