"""Fused covariance build + Cholesky + L^-1 (pg_build_potrf_trtri) with and without the recursive top split (PG_REC_MIN is read
once per process: run this twice, e.g. PG_REC_MIN=0 and default), timed with events, with three checks that do not need a
reference run: |L Minv - I| on sampled columns, K alpha = y through the factor, and log det."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
for n in ([int(a) for a in sys.argv[1:]] or [12288, 16384]):
    d = 8
    rng = np.random.default_rng(1234)
    x = torch.from_numpy(rng.random((n, d))).cuda()
    hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
    spec = make_spec([0], [0], [d + 1])
    a = ops.empty(n, n); m = ops.empty(n, n)
    invd = ops.potrf_workspace(n, torch.float64); info = torch.zeros(1, dtype=torch.int32, device="cuda")
    def run():
        ops.build_factor(spec, hp, x, a, invd, info, m)
    run(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    if os.environ.get('PROBE_NOCHECK'):
        print(f'n={n} min {min(ts):.2f} ms'); continue
    L = torch.tril(a); M = torch.tril(m)
    cols = torch.arange(0, n, 997, device="cuda")
    E = L @ M[:, cols]; E[cols, torch.arange(len(cols), device="cuda")] -= 1.0
    k = ops.empty(n, n); ops.kernel_build(spec, hp, x, None, k, lower_only=False, jitter=1e-7)
    v = torch.from_numpy(rng.standard_normal(n)).cuda()
    r = L @ (L.T @ v) - k @ v
    print(f"n={n} REC_MIN={os.environ.get('PG_REC_MIN','default')} build+potrf+trtri min {min(ts):.2f} ms med {sorted(ts)[2]:.2f} info={int(info.item())} "
          f"|L M - I|max={E.abs().max().item():.2e} |LL^T v - K v|/|Kv|={(r.norm()/(k@v).norm()).item():.2e} "
          f"logdet={2*torch.log(torch.diagonal(L)).sum().item():.10f} coupled panels {ops.last_coupled_panels()}", flush=True)
    del a, m, L, M, k
