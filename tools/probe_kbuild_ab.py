"""A/B of pg_kernel_build between two builds of the library on the same box (raw ctypes: only pg_create / pg_kernel_build are used).
python tools/probe_kbuild_ab.py libA.so libB.so"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._lib import CovSpec
n = 16384
vp, i_, l_, d_ = C.c_void_p, C.c_int, C.c_long, C.c_double
def load(path):
    lib = C.CDLL(path)
    lib.pg_create.argtypes = [C.POINTER(vp)]
    lib.pg_kernel_build.argtypes = [vp, i_, C.POINTER(CovSpec), vp, vp, l_, i_, vp, l_, i_, i_, i_, i_, d_, vp, l_, i_, i_, vp]
    h = vp(); assert lib.pg_create(C.byref(h)) == 0
    return lib, h
libs = [(p, *load(p)) for p in sys.argv[1:]]
k = torch.empty(n, n, dtype=torch.float64, device="cuda")
for d in (8, 2):
    x = torch.from_numpy(np.random.default_rng(d).random((n, d))).cuda()
    hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
    sp = CovSpec(); sp.ncomp = 1; sp.kind[0] = 0; sp.off[0] = 0; sp.nnoise = 1; sp.noise_off[0] = d + 1
    for rep in range(2):
        for path, lib, h in libs:
            res = []
            for lower in (1, 0):
                def run():
                    rc = lib.pg_kernel_build(h, 0, C.byref(sp), vp(hp.data_ptr()), vp(x.data_ptr()), d, n, vp(0), 0, n, d, lower, 0, 1e-7, vp(k.data_ptr()), n, n, n,
                                             vp(torch.cuda.current_stream().cuda_stream))
                    assert rc == 0
                run(); torch.cuda.synchronize(); best = 1e9
                for _ in range(5):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); run(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
                res.append(best)
            print("d=%d %-40s lower %.3f ms  full %.3f ms (%.0f GB/s)" % (d, os.path.basename(path), res[0], res[1], 8 * n * n / res[1] / 1e6), flush=True)
