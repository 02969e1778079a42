"""Inputs of the MFMA-utilisation counter pass (north_star: "MFMA utilisation on the O(N^3) factorise against gfx950 peak"):
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d out -- python3 tools/probe_mfma_pass.py potrf|lauum|fused
one warm-up call, then ONE measured call at N = 16384 (a counter-collecting profiler runs one kernel at a time: pg_create's probe
keeps the classic chain there, as in the FETCH / WRITE passes).  tools/pmc_mfma_summary.py reads the result."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
what = sys.argv[1] if len(sys.argv) > 1 else "potrf"
n, d = (int(sys.argv[2]) if len(sys.argv) > 2 else 16384), 8
rng = np.random.default_rng(1234)
x = torch.from_numpy(rng.random((n, d))).cuda()
hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
spec = make_spec([0], [0], [d + 1])
a = ops.empty(n, n); invd = ops.potrf_workspace(n, torch.float64); info = torch.zeros(1, dtype=torch.int32, device="cuda")
m = ops.empty(n, n) if what != "potrf" else None
for it in range(2):
    ops.kernel_build(spec, hp, x, None, a, lower_only=True, jitter=1e-7)       # marker: the measured call follows the LAST build
    if what == "potrf":
        ops.potrf(a, invd, info)
    elif what == "fused":
        ops.potrf_trtri(a, invd, info, m)
    else:
        ops.potrf_trtri(a, invd, info, m)
        torch.cuda.synchronize()
        ops.lauum(m, a)
    torch.cuda.synchronize()
print(what, "info", int(info.item()), "coupled chain enabled:", ops.coupled_chain(), "panels coupled:", ops.last_coupled_panels())
