"""Ad-hoc at-scale timing of the O(n^3) pieces (not part of the test suite)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pygpr_amd._ops import get_ops, make_spec, pad_to
from pygpr_amd._lib import GEMM_NT
ops = get_ops()
def ev(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
for n, d in [(8192, 8), (16384, 8)]:
    rng = np.random.default_rng(1234)
    x = torch.from_numpy(rng.random((n, d))).cuda()
    y = torch.from_numpy(np.sin(-rng.random(n))).cuda()
    hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
    spec = make_spec([0], [0], [d + 1])
    k = ops.empty(n, n); kl = ops.empty(n, n)
    t = ev(lambda: ops.kernel_build(spec, hp, x, None, k, jitter=1e-7))
    print(f"n={n} kbuild full {t:.3f} ms  {8*n*n/t/1e6:.0f} GB/s")
    t = ev(lambda: ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7))
    print(f"n={n} kbuild lower {t:.3f} ms")
    invd = ops.potrf_workspace(n, torch.float64); info = torch.zeros(1, dtype=torch.int32, device="cuda")
    def fac():
        kl.copy_(k); ops.potrf(kl, invd, info)
    tc = ev(lambda: kl.copy_(k))
    t = ev(fac) - tc
    print(f"n={n} potrf {t:.2f} ms  {n**3/3/t/1e9:.1f} TF/s  info={int(info.item())}")
    minv = ops.zeros(n, n)
    t = ev(lambda: ops.trtri(kl, invd, minv)); print(f"n={n} trtri {t:.2f} ms  {n**3/3/t/1e9:.1f} TF/s")
    kinv = ops.zeros(n, n)
    t = ev(lambda: ops.lauum(minv, kinv)); print(f"n={n} lauum {t:.2f} ms  {n**3/3/t/1e9:.1f} TF/s")
    alpha = ops.empty(n)
    t = ev(lambda: ops.potrs_vec(kl, invd, y, alpha)); print(f"n={n} potrs_vec {t:.2f} ms")
    out = ops.zeros(2 + d + 1); work = ops.empty(ops.nlml_grad_worksize(n, d + 2))
    t = ev(lambda: ops.nlml_grad(spec, hp, x, n, kinv, alpha, out[1:], work)); print(f"n={n} nlml_grad {t:.3f} ms")
    # big SYRK alone
    p = k[:, :256].contiguous()
    t = ev(lambda: ops.gemm_raw(GEMM_NT, n, n, 256, -1.0, p, p, 1.0, kinv, tri=1))
    print(f"n={n} syrk K=256 lower {t:.3f} ms  {n*(n+128)*256/t/1e9:.1f} TF/s")
    p2 = k[:, :2048].contiguous()
    t = ev(lambda: ops.gemm_raw(GEMM_NT, n, n, 2048, -1.0, p2, p2, 1.0, kinv, tri=1))
    print(f"n={n} syrk K=2048 lower {t:.3f} ms  {n*(n+128)*2048/t/1e9:.1f} TF/s")
    del k, kl, minv, kinv
