"""Does a process that has used RCCL (torch.distributed, backend nccl) still run the look-ahead factorisation at full speed?
One rank on one GPU; PG_PROBE_NCCL=0 skips the process group for the comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
side = os.environ.get("PG_PROBE_SIDE", "0") == "1"
if side:
    torch.cuda.set_stream(torch.cuda.Stream())
if os.environ.get("PG_PROBE_NCCL", "1") == "1":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.ones(16, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    print("nccl up", float(t[0]))
for n in (8192, 16384):
    d = 8
    rng = np.random.default_rng(1234)
    x = torch.from_numpy(rng.random((n, d))).cuda()
    hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
    spec = make_spec([0], [0], [d + 1])
    kl = ops.empty(n, n); minv = ops.empty(n, n)
    invd = ops.potrf_workspace(n, torch.float64); info = torch.zeros(1, dtype=torch.int32, device="cuda")
    def plain():
        ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7); ops.potrf(kl, invd, info)
    def fused():
        ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7); ops.potrf_trtri(kl, invd, info, minv)
    for name, fn in (("plain", plain), ("fused", fused)):
        fn(); torch.cuda.synchronize(); best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
        if os.environ.get("PG_PROBE_NCCL", "1") == "1":
            t = torch.ones(16, device="cuda"); dist.all_reduce(t)
        print(f"n={n} {name}: {best:.2f} ms coupled_panels={ops.last_coupled_panels()} side={side}", flush=True)
    del kl, minv
if dist.is_initialized():
    dist.destroy_process_group()
