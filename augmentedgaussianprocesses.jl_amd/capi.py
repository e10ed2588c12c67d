"""ctypes binding of include/agp_hip.h (libagp_hip.so).

This is the same boundary a Julia `ccall` shim binds; nothing here computes.
Loading fails loudly when the HIP library is missing: there is NO CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AGP_HIP_LIB: an alternative build of the same library (kernel experiments are compared as two .so files in one GPU session)
LIB_PATH = os.environ.get("AGP_HIP_LIB") or os.path.join(_HERE, "libagp_hip.so")

AGP_OK = 0
ERR_NAMES = {
    1: "AGP_ERR_INVALID",
    2: "AGP_ERR_NOT_POSDEF",
    3: "AGP_ERR_NEG_KTILDE",
    4: "AGP_ERR_BAD_BATCH",
    5: "AGP_ERR_UNSUPPORTED",
    6: "AGP_ERR_LABELS",
    7: "AGP_ERR_HIP",
    8: "AGP_ERR_NOMEM",
}
F64, F32 = 0, 1
K_SQEXP, K_MATERN52, K_MATERN32, K_EXPONENTIAL = 0, 1, 2, 3
LIK_GAUSSIAN, LIK_LOGISTIC, LIK_STUDENTT, LIK_LOGISTICSOFTMAX, LIK_MULTIOUTPUT = 0, 1, 2, 3, 4
LIK_LAPLACE, LIK_BAYESIANSVM, LIK_POISSON, LIK_NEGBINOMIAL, LIK_HETEROSCEDASTIC = 5, 6, 7, 8, 9
OPT_ADAM, OPT_DESCENT, OPT_MOMENTUM = 0, 1, 2  # agp_svgp_hyper_rule
ELBO_CORRECTED, ELBO_REFERENCE = 0, 1
FLAG_STALE_K = 1  # reference_compat_stale_K
SHARD_LATENT, SHARD_BATCH = 0, 1
COMM_ID_BYTES = 128
# int32_t (*agp_allreduce_fn)(void* user, void* buf, int64_t count, int32_t dtype, void* hip_stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)
MAT_L, MAT_KINV, MAT_KNM, MAT_KAPPA, VEC_KTILDE, VEC_MEAN_F, VEC_VAR_F, VEC_THETA, VEC_C, VEC_GAMMA, VEC_ALPHA = range(11)


class KernelDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("ard", C.c_int32),
        ("variance", C.c_double),
        ("scale", C.c_double),
        ("ard_scales_host", C.POINTER(C.c_double)),
        ("has_variance", C.c_int32),   # the kernel object is `sigma2 * k`  (what update_kernel! may step)
        ("has_transform", C.c_int32),  # the kernel object is `k o ScaleTransform / ARDTransform`
    ]


class LikDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_class", C.c_int32), ("p0", C.c_double), ("p1", C.c_double)]


class SvgpDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("n_latent", C.c_int32),
        ("latent_offset", C.c_int32),
        ("stochastic", C.c_int32),
        ("m", C.c_int64),
        ("D", C.c_int64),
        ("max_batch", C.c_int64),
        ("lik", LikDesc),
        ("jitter", C.c_double),
        ("rm_kappa", C.c_double),
        ("rm_tau", C.c_double),
        ("elbo_mode", C.c_int32),
        ("flags", C.c_int32),
    ]


# every symbol include/agp_hip.h declares: name -> (restype, argtypes)
_VP, _I32, _I64, _DBL = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_PI32, _PI64, _PDBL, _PVP = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_void_p)
_PK = C.POINTER(KernelDesc)
SYMBOLS = {
    "agp_version": (_I32, []),
    "agp_ctx_create": (_I32, [_I32, _VP, _PVP]),
    "agp_ctx_destroy": (_I32, [_VP]),
    "agp_ctx_sync": (_I32, [_VP]),
    "agp_ctx_task_graph_fallbacks": (_I32, [_VP, _PI64]),
    "agp_last_error": (C.c_char_p, [_VP]),
    "agp_kernelmatrix": (_I32, [_VP, _I32, _PK, _VP, _I64, _I64, _VP, _VP, _I64, _I64, _I64, _VP, _I64]),
    "agp_potrf_jitter": (_I32, [_VP, _I32, _VP, _I64, _I64, _DBL, _PI32]),
    "agp_spd_inverse": (_I32, [_VP, _I32, _VP, _I64, _I64, _VP, _I64, _PDBL, _PI32]),
    "agp_solve_right_spd": (_I32, [_VP, _I32, _VP, _I64, _I64, _VP, _I64, _I64, _VP, _I64, _PI32]),
    "agp_mfma_peak": (_I32, [_VP, _I32, _PDBL]),
    "agp_svgp_create": (_I32, [_VP, C.POINTER(SvgpDesc), _PVP]),
    "agp_svgp_destroy": (_I32, [_VP]),
    "agp_svgp_set_kernel": (_I32, [_VP, _I32, _PK]),
    "agp_svgp_set_Z": (_I32, [_VP, _I32, _VP, _I64]),
    "agp_svgp_get_Z": (_I32, [_VP, _I32, _VP, _I64]),
    "agp_svgp_set_prior_mean": (_I32, [_VP, _I32, _VP]),
    "agp_svgp_refresh_K": (_I32, [_VP]),
    "agp_svgp_set_opt_state": (_I32, [_VP, _I64]),
    "agp_svgp_get_opt_state": (_I32, [_VP, _PI64]),
    "agp_svgp_cavi_step": (_I32, [_VP, _VP, _I64, _VP, _VP, _I64, _DBL]),
    "agp_svgp_step_counters": (_I32, [_VP, _PI64, _PI64]),
    "agp_svgp_hyper_counters": (_I32, [_VP, _PI64, _PI64]),
    "agp_svgp_step_local": (_I32, [_VP, _VP, _I64, _VP, _VP, _I64, _DBL]),
    "agp_svgp_hyper_configure": (_I32, [_VP, _I32, _DBL, _I32, _DBL, _DBL, _DBL, _DBL]),
    "agp_svgp_hyper_rule": (_I32, [_VP, _I32, _DBL, _I32, _DBL]),
    "agp_svgp_hypergrad": (_I32, [_VP, _I32, _PDBL, _PDBL, _VP]),
    "agp_svgp_hyper_step": (_I32, [_VP]),
    "agp_svgp_get_kernel": (_I32, [_VP, _I32, _PDBL, _PDBL]),
    "agp_svgp_set_multioutput": (_I32, [_VP, _I32, C.POINTER(LikDesc), _PDBL, _DBL, _DBL, _DBL, _DBL]),
    "agp_svgp_get_A": (_I32, [_VP, _PDBL]),
    "agp_svgp_elbo_terms": (_I32, [_VP, _PDBL]),
    "agp_svgp_set_batch_shard": (_I32, [_VP, _I32, _I32]),
    "agp_svgp_mo_shard": (_I32, [_VP, _I32]),
    "agp_svgp_mo_fbuf_ptr": (_I32, [_VP, _PVP, _PI64]),
    "agp_svgp_mo_mix": (_I32, [_VP]),
    "agp_svgp_mo_refresh_f": (_I32, [_VP]),
    "agp_svgp_mo_predict_from_f": (_I32, [_VP, _I64, _I32, _VP, _VP, _PDBL, _PDBL, _I32]),
    "agp_svgp_prefetch": (_I32, [_VP, _VP, _I64, _VP, _I64]),
    "agp_svgp_lsm_gamma": (_I32, [_VP]),
    "agp_svgp_lsm_alpha": (_I32, [_VP]),
    "agp_svgp_lsm_gsum_ptr": (_I32, [_VP, _PVP, _PI64]),
    "agp_svgp_step_stats": (_I32, [_VP]),
    "agp_svgp_stats_ptr": (_I32, [_VP, _PVP, _PI64]),
    "agp_svgp_step_global": (_I32, [_VP]),
    "agp_svgp_timing_enable": (_I32, [_VP, _I32]),
    "agp_svgp_timing_read": (_I32, [_VP, _PI64, _PDBL]),
    "agp_svgp_check_status": (_I32, [_VP]),
    "agp_svgp_elbo": (_I32, [_VP, _VP, _I64, _VP, _VP, _I64, _DBL, _I32, _PDBL]),
    "agp_svgp_elbo_enqueue": (_I32, [_VP, _VP, _I64, _VP, _VP, _I64, _DBL, _I32, _PI32]),
    "agp_svgp_elbo_fetch": (_I32, [_VP, _I32, _I32, _PDBL, _PI32]),
    "agp_svgp_get_state": (_I32, [_VP, _I32, _VP, _VP, _VP, _VP]),
    "agp_svgp_set_state": (_I32, [_VP, _I32, _VP, _VP]),
    "agp_svgp_get_matrix": (_I32, [_VP, _I32, _I32, _VP, _I64, _I64]),
    "agp_svgp_last_batch": (_I32, [_VP, _PI64]),
    "agp_svgp_invalidate_data": (_I32, [_VP]),
    "agp_svgp_init_state": (_I32, [_VP]),
    "agp_svgp_predict_f_cov": (_I32, [_VP, _VP, _I64, _I64, _VP, _VP]),
    "agp_comm_unique_id": (_I32, [C.POINTER(C.c_uint8)]),
    "agp_comm_init": (_I32, [_VP, _I32, _I32, C.POINTER(C.c_uint8), _PVP]),
    "agp_comm_init_callback": (_I32, [_VP, _I32, _I32, ALLREDUCE_FN, _VP, _PVP]),
    "agp_comm_destroy": (_I32, [_VP]),
    "agp_comm_info": (_I32, [_VP, _PI32, _PI32, _PI32]),
    "agp_comm_allreduce": (_I32, [_VP, _VP, _I64, _I32]),
    "agp_comm_timing": (_I32, [_VP, _I32]),
    "agp_comm_stats": (_I32, [_VP, _PI64, _PI64, _PDBL]),
    "agp_comm_standin_allreduce": (_I32, [_VP, _I64, _I32, _VP, _I32, _I32, _DBL, _VP]),
    "agp_svgp_cavi_step_multi": (_I32, [_VP, _VP, _I32, _VP, _I64, _VP, _VP, _I64, _DBL]),
    "agp_svgp_elbo_multi": (_I32, [_VP, _VP, _I32, _PDBL]),
    "agp_svgp_hyper_step_multi": (_I32, [_VP, _VP, _I32]),
    "agp_svgp_predict_multi": (_I32, [_VP, _VP, _I32, _VP, _I64, _I64, _VP, _VP, _PDBL, _PDBL, _I32]),
    "agp_svgp_predict_f": (_I32, [_VP, _VP, _I64, _I64, _VP, _VP]),
    "agp_svgp_predict_y": (_I32, [_VP, _VP, _I64, _I64, _VP]),
    "agp_svgp_proba_y": (_I32, [_VP, _VP, _I64, _I64, _PDBL, _PDBL, _I32, _VP, _VP]),
    "agp_nearest_center": (_I32, [_VP, _I32, _VP, _I64, _I64, _I64, _VP, _I64, _I64, _VP, _VP]),
    "agp_kmeans": (_I32, [_VP, _I32, _VP, _I64, _I64, _I64, _VP, _I64, _I64, _I32, _DBL, _VP, _VP, _PI32, _PDBL, _PI32]),
    "agp_svgp_online_snapshot": (_I32, [_VP, _I32, _VP, _I64, _VP, _PDBL]),
    "agp_svgp_set_online_prior": (_I32, [_VP, _I32, _VP, _I64, _I64, _VP, _I64, _VP, _DBL]),
    "agp_svgp_adopt_local": (_I32, [_VP, _VP]),
    "agp_svgp_online_first_step": (_I32, [_VP, _VP, _VP, _I64, _VP, _I64]),
    "agp_svgp_hyper_apply": (_I32, [_VP, _I32, _PDBL, _PDBL, _VP]),
    "agp_svgp_hyper_opt_state": (_I32, [_VP, _I32, _I32, _PDBL, _PDBL, _PI32]),
    "agp_svgp_set_quadrature": (_I32, [_VP, _PDBL, _PDBL, _I32]),
    "agp_svgp_get_lik_param": (_I32, [_VP, _PDBL]),
    "agp_svgp_set_lsm_alpha": (_I32, [_VP, _VP, _I64]),
    "agp_svgp_set_lik_param": (_I32, [_VP, _DBL]),
}

_lib = None


class AGPError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__(f"{ERR_NAMES.get(status, status)}: {msg}")


def lib():
    """Load libagp_hip.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
            )
        # torch (device memory / streams / RCCL plumbing) bundles its own libamdhip64: import it FIRST so that this
        # library binds to the same HIP runtime instance instead of loading a second one from /opt/rocm.
        import torch  # noqa: F401

        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(ctx, status):
    if status != AGP_OK:
        msg = lib().agp_last_error(ctx)
        raise AGPError(status, msg.decode() if msg else "")
