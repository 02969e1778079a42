"""ctypes binding of libpygpr_hip.so (C ABI: include/pygpr_hip.h).

There is no CPU fallback: `load()` raises if the shared object is missing, and every compute call
needs a HIP device.  Build the library with `python __graft_entry__.py` (hipcc, gfx950).
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpygpr_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "pygpr_hip.h")

PG_F64, PG_F32 = 0, 1
PG_KIND_RBF, PG_KIND_MATERN52, PG_KIND_SQDIST = 0, 1, 2
PG_MAX_COMP, PG_MAX_DIM = 4, 64
PAD = 256  # every dimension given to the O(n^3) entry points is a multiple of this

GEMM_NT, GEMM_NT_RP, GEMM_NN, GEMM_TN, GEMM_TT, GEMM_NT_64, GEMM_NT_64x128, GEMM_NT_32x64, GEMM_NT_32x128, GEMM_TT_64, GEMM_NT_32x32 = 0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11
GEMM_TN_64 = 13


class CovSpec(C.Structure):
    _fields_ = [
        ("ncomp", C.c_int),
        ("kind", C.c_int * PG_MAX_COMP),
        ("off", C.c_int * PG_MAX_COMP),
        ("nnoise", C.c_int),
        ("noise_off", C.c_int * PG_MAX_COMP),
    ]


_vp, _i, _l, _d = C.c_void_p, C.c_int, C.c_long, C.c_double
_SIGS = {
    "pg_version": (C.c_int, []),
    "pg_last_error": (C.c_char_p, []),
    "pg_create": (_i, [C.POINTER(_vp)]),
    "pg_destroy": (_i, [_vp]),
    "pg_kernel_build": (_i, [_vp, _i, C.POINTER(CovSpec), _vp, _vp, _l, _i, _vp, _l, _i, _i, _i, _i, _d, _vp, _l, _i, _i, _vp]),
    "pg_kernel_build_batched": (_i, [_vp, _i, C.POINTER(CovSpec), _vp, _l, _vp, _l, _l, _i, _vp, _l, _l, _i, _i, _i, _d, _vp, _l, _l, _i, _i, _i, _vp]),
    "pg_predict_mean_q_kt_batched": (_i, [_vp, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _vp, _l, _vp, _l, C.POINTER(CovSpec), _vp, _l, _vp, _l,
                                          _i, _vp]),
    "pg_grbcm_local_terms_batched": (_i, [_vp, _i, _i, _vp, _l, _vp, _l, _vp, _i, _i, _i, _vp, _l, _vp, _vp, _l, _vp]),
    "pg_kernel_grad_build": (_i, [_vp, _i, C.POINTER(CovSpec), _vp, _vp, _l, _i, _i, _vp, _vp]),
    "pg_potrf_worksize": (_l, [_i, _i]),
    "pg_potrf": (_i, [_vp, _i, _i, _vp, _l, _vp, _vp, _vp]),
    "pg_potrf_trtri": (_i, [_vp, _i, _i, _vp, _l, _vp, _vp, _vp, _l, _vp]),
    "pg_potrs_vec_worksize": (_l, [_i, _i]),
    "pg_potrs_vec": (_i, [_vp, _i, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp]),
    "pg_trtri": (_i, [_vp, _i, _i, _vp, _l, _vp, _vp, _l, _vp]),
    "pg_lauum": (_i, [_vp, _i, _i, _vp, _l, _vp, _l, _vp]),
    "pg_potri": (_i, [_vp, _i, _i, _vp, _l, _vp, _vp, _l, _vp, _vp]),
    "pg_logdet": (_i, [_vp, _i, _i, _vp, _l, _vp, _vp]),
    "pg_trmv": (_i, [_vp, _i, _i, _vp, _l, _i, _vp, _vp, _vp, _vp]),
    "pg_alpha_nlml_batched": (_i, [_vp, _i, _i, _i, _vp, _l, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _i, _vp]),
    "pg_lauum_batched": (_i, [_vp, _i, _i, _vp, _l, _l, _vp, _l, _l, _i, _vp]),
    "pg_nlml_grad_batched": (_i, [_vp, _i, C.POINTER(CovSpec), _vp, _l, _vp, _l, _l, _i, _i, _vp, _l, _l, _vp, _l, _vp, _l, _i, _vp, _l, _i, _vp]),
    "pg_alpha_nlml_async": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pg_nlml_value": (_i, [_vp, _i, _i, _vp, _l, _vp, _vp, _vp, _vp]),
    "pg_nlml_grad_worksize": (_l, [_i, _i]),
    "pg_nlml_grad": (_i, [_vp, _i, C.POINTER(CovSpec), _vp, _vp, _l, _i, _i, _vp, _l, _vp, _vp, _i, _vp, _l, _vp]),
    "pg_predict_mean_q": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _vp, _d, _vp, _vp]),
    "pg_predict_mean_q_kt": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _vp, _d, _vp, _vp]),
    "pg_trmm_lower": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp]),
    "pg_syrk_tn_sub": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _l, _i, _vp]),
    "pg_trmm_lower_kt_batched": (_i, [_vp, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _i, _vp]),
    "pg_syrk_nt_sub_batched": (_i, [_vp, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _i, _i, _vp]),
    "pg_grbcm_local_terms": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _l, _vp, _vp, _vp]),
    "pg_grbcm_finish": (_i, [_vp, _i, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pg_grbcm_weighted_prec": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _vp, _l, _i, _vp]),
    "pg_symmetrize": (_i, [_vp, _i, _i, _vp, _l, _vp]),
    "pg_grbcm_finish_full": (_i, [_vp, _i, _i, _vp, _l, _vp, _vp, _vp, _l, _vp, _vp]),
    "pg_sqdist_argmin": (_i, [_vp, _i, _vp, _l, _i, _vp, _l, _i, _i, _vp, _l, _vp, _vp]),
    "pg_tril": (_i, [_vp, _i, _i, _vp, _l, _vp]),
    "pg_set_lookahead": (_i, [_vp, _i]),
    "pg_set_outer_panel": (_i, [_vp, _i]),
    "pg_set_recursive_split": (_i, [_vp, _i]),
    "pg_profile": (_i, [_vp, _i]),
    "pg_profile_read": (_i, [_vp, C.POINTER(_d), C.POINTER(_d), C.POINTER(_l)]),
    "pg_build_potrf_trtri": (_i, [_vp, _i, C.POINTER(CovSpec), _vp, _vp, _l, _i, _i, _d, _vp, _l, _i, _vp, _vp, _vp, _l, _vp]),
    "pg_build_potrf_trtri_checked": (_i, [_vp, _i, C.POINTER(CovSpec), _vp, _vp, _l, _i, _i, _d, _vp, _l, _i, _vp, _vp, _vp, _l, _vp,
                                          C.POINTER(_i)]),
    "pg_build_potrf_trtri_batched": (_i, [_vp, _i, C.POINTER(CovSpec), _vp, _l, _vp, _l, _l, _i, _i, _d, _vp, _l, _l, _i, _vp, _l, _vp, _vp,
                                          _l, _l, _i, _vp]),
    "pg_alpha_batched": (_i, [_vp, _i, _i, _vp, _l, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _i, _vp]),
    "pg_potrs_worksize": (_l, [_i, _i, _i, _i]),
    "pg_potrs": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _vp, _l, _vp, _l, _vp, _l, _vp, _vp]),
    "pg_trsm_lower": (_i, [_vp, _i, _i, _i, _vp, _l, _vp, _vp, _l, _vp, _l, _vp, _l, _vp, _vp]),
    "pg_set_spin_budget": (_i, [_vp, _l]),
    "pg_set_rearm_after": (_i, [_vp, _i]),
    "pg_chain_rearms": (_i, [_vp]),
    "pg_wait_budget_us": (_l, [_vp, _i]),
    "pg_spin_probe": (_i, [_vp, _i, _vp, _vp]),
    "pg_chain_timeouts": (_i, [_vp]),
    "pg_last_coupled_panels": (_i, [_vp]),
    "pg_last_deferred_panels": (_i, [_vp]),
    "pg_set_deferred_block": (_i, [_vp, _i]),
    "pg_set_coupled_chain": (_i, [_vp, _i]),
    "pg_coupled_chain": (_i, [_vp]),
    "pg_leaf_raw": (_i, [_vp, _i, _vp, _l, _vp, _l, _vp, _i, _vp]),
    "pg_rowstep_raw": (_i, [_vp, _i, _i, _vp, _l, _i, _i, _vp, _vp, _vp, _vp]),
    "pg_gemm_raw": (_i, [_vp, _i, _i, _i, _i, _i, _d, _vp, _l, _vp, _l, _d, _vp, _l, _i, _i, _i, _vp]),
}

_lib = None


def header_symbols():
    """Names of every function declared in include/pygpr_hip.h."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pg_[a-z_0-9]+)\s*\(", text)))


def load(check_symbols=False):
    """dlopen the library (no GPU needed for that) and attach the prototypes."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libpygpr_hip.so not found at %s -- the HIP extension is required (no CPU fallback); "
                "build it with `python __graft_entry__.py`" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    if check_symbols:
        missing = [s for s in header_symbols() if not hasattr(_lib, s)]
        unbound = [s for s in header_symbols() if s not in _SIGS]
        if missing or unbound:
            raise RuntimeError("C ABI mismatch: missing in .so %s, unbound in _lib.py %s" % (missing, unbound))
    return _lib


def last_error():
    return load().pg_last_error().decode()


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("libpygpr_hip %s failed (rc=%d): %s" % (what, rc, last_error()))


def build_id():
    """Identity of the running build: hash of the kernel sources + C header (what a profile must match to describe
    this build) and of the shared object itself."""
    import hashlib

    csrc = os.path.join(_HERE, "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)) + [HEADER]:
        path = f if os.path.isabs(f) else os.path.join(csrc, f)
        if path.endswith((".hip", ".h")):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    lib = hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()[:16] if os.path.exists(LIB_PATH) else None
    return {"src_sha16": h.hexdigest()[:16], "lib_sha16": lib}
