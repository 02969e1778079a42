"""build + potrf time (look-ahead on), best of 5, for the sizes on the command line: one line per size.  For same-box A/B of schedule switches
(environment variables are read once per process: run one process per setting)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
tag = os.environ.get("PG_TAG", "")
for n in ([int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]):
    d = 8
    rng = np.random.default_rng(1234)
    x = torch.from_numpy(rng.random((n, d))).cuda()
    hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
    spec = make_spec([0], [0], [d + 1])
    kl = ops.empty(n, n)
    invd = ops.potrf_workspace(n, torch.float64); info = torch.zeros(1, dtype=torch.int32, device="cuda")
    def run():
        ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7); ops.potrf(kl, invd, info)
    run(); run(); torch.cuda.synchronize(); ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"[{tag}] n={n} build+potrf best {ts[0]:.3f} median {ts[3]:.3f} ms info={int(info.item())} coupled_panels={ops.last_coupled_panels()}", flush=True)
    del kl, invd
