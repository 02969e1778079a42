import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
if os.environ.get('PG_TORCH_SIDE_STREAM'):
    _side = torch.cuda.Stream(); torch.cuda.set_stream(_side)   # everything below runs on a non-default stream
for n in ([int(a) for a in sys.argv[1:]] or [8192, 16384]):
    d = 8
    rng = np.random.default_rng(1234)
    x = torch.from_numpy(rng.random((n, d))).cuda()
    hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
    spec = make_spec([0], [0], [d + 1])
    kl = ops.empty(n, n)
    invd = ops.potrf_workspace(n, torch.float64); info = torch.zeros(1, dtype=torch.int32, device="cuda")
    def run():
        ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7); ops.potrf(kl, invd, info)
    for la in (1, 0):
        ops.set_lookahead(la)
        run(); torch.cuda.synchronize(); best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
        print(f"n={n} lookahead={la} build+potrf {best:.2f} ms -> potrf ~{n**3/3/(best-0.45*(n/16384)**2)/1e9:.1f} TF/s info={int(info.item())}", flush=True)
    minv = ops.empty(n, n)
    def fused():
        ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7); ops.potrf_trtri(kl, invd, info, minv)
    def sep():
        ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7); ops.potrf(kl, invd, info); ops.trtri(kl, invd, minv)
    ops.set_lookahead(1)
    for name, fn in (("separate", sep), ("fused", fused)):
        fn(); torch.cuda.synchronize(); best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
        print(f"n={n} build+potrf+trtri {name}: {best:.2f} ms", flush=True)
    del minv
