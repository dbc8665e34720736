"""ctypes binding of libbayesic_hip.so (C ABI: include/bayesic_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails this
module raises.  Device memory and streams come from torch (plumbing only); the
library itself has no torch dependency -- every entry point takes raw device
pointers.
"""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t,
                    c_uint32, c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libbayesic_hip.so")


class BayesicHipError(RuntimeError):
    """A libbayesic_hip.so call returned a non-zero status."""


# name -> (restype, argtypes); every symbol declared in include/bayesic_hip.h.
SIGNATURES = {
    "bsc_ctx_create": (c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    "bsc_ctx_destroy": (c_int, [c_void_p]),
    "bsc_ctx_set_stream": (c_int, [c_void_p, c_void_p]),
    "bsc_ctx_reserve": (c_int, [c_void_p, c_size_t]),
    "bsc_ctx_sync": (c_int, [c_void_p]),
    "bsc_capture_begin": (c_int, [c_void_p]),
    "bsc_capture_end": (c_int, [c_void_p, POINTER(c_void_p)]),
    "bsc_graph_launch": (c_int, [c_void_p, c_void_p]),
    "bsc_graph_destroy": (c_int, [c_void_p]),
    "bsc_ctx_set_mfma_split": (c_int, [c_void_p, c_int]),
    "bsc_ctx_set_option": (c_int, [c_void_p, c_char_p, c_int64]),
    "bsc_ctx_get_option": (c_int, [c_void_p, c_char_p, POINTER(c_int64)]),
    "bsc_ctx_option_name": (c_int, [c_int32, POINTER(c_char_p)]),
    "bsc_ctx_profile": (c_int, [c_void_p, c_int]),
    "bsc_ctx_profile_read": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int64)]),
    "bsc_ctx_profile_read_slot": (c_int, [c_void_p, c_int, POINTER(c_double), POINTER(c_int64)]),
    "bsc_comm_unique_id": (c_int, [c_void_p]),
    "bsc_comm_init_rank": (c_int, [c_void_p, c_void_p, c_int32, c_int32]),
    "bsc_comm_destroy": (c_int, [c_void_p]),
    "bsc_comm_info": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "bsc_allreduce_sum": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    "bsc_allreduce_max": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    "bsc_allreduce_sum_begin": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int32]),
    "bsc_allreduce_sum_end": (c_int, [c_void_p, c_int32]),
    "bsc_natgrad_update_f32_2d": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_int64, c_int64, c_int64,
                                          c_float, c_float, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "bsc_lda_sstats_round_columns": (c_int, [c_void_p, c_int32, POINTER(c_int64)]),
    "bsc_device_info": (c_int, [c_void_p, POINTER(c_int64)]),
    "bsc_last_error": (c_char_p, []),
    "bsc_version": (c_int, []),
    "bsc_malloc": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "bsc_free": (c_int, [c_void_p, c_void_p]),
    "bsc_h2d": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "bsc_d2h": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "bsc_memset": (c_int, [c_void_p, c_void_p, c_int, c_size_t]),
    "bsc_event_create": (c_int, [POINTER(c_void_p)]),
    "bsc_event_destroy": (c_int, [c_void_p]),
    "bsc_event_record": (c_int, [c_void_p, c_void_p]),
    "bsc_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "bsc_philox_normal": (c_int, [c_void_p, c_uint64, c_uint32, c_uint32, c_int32, c_int32,
                                  c_void_p]),
    "bsc_blr_sample": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_uint64, c_uint32,
                               c_void_p, c_void_p, c_void_p]),
    "bsc_blr_data_pass": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32,
                                  c_void_p, c_int32, c_void_p, c_void_p]),
    "bsc_blr_data_pass_partial": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32,
                                          c_void_p, c_int32]),
    "bsc_blr_data_pass_sweep": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32,
                                        c_void_p, c_int32, c_void_p, c_void_p, c_int32]),
    "bsc_blr_data_pass_partial_sweep": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                                c_int32, c_void_p, c_int32, c_int32]),
    "bsc_blr_read_stamps": (c_int, [c_void_p, c_void_p, c_int32, POINTER(c_int32)]),
    "bsc_blr_pass_count": (c_int, [c_void_p, c_void_p, c_int32, c_int32, POINTER(c_int32)]),
    "bsc_blr_fused_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_double,
                                     c_double, c_double, c_double, c_int64, c_double, c_double,
                                     c_double, c_double, c_uint64, c_uint32, c_void_p, c_int32,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "bsc_blr_fused_update_general": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_double,
                                             c_double, c_double, c_double, c_double, c_int64, c_double, c_double,
                                             c_double, c_double, c_uint64, c_uint32, c_void_p, c_int32,
                                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "bsc_blr_pass_update": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                    c_double, c_double, c_double, c_double, c_int64, c_double, c_double, c_double,
                                    c_double, c_uint64, c_uint32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "bsc_blr_pass_update_general": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                            c_double, c_double, c_double, c_double, c_double, c_int64, c_double, c_double,
                                            c_double, c_double, c_uint64, c_uint32, c_void_p, c_int32, c_void_p, c_void_p,
                                            c_void_p, c_void_p]),
    "bsc_blr_noise": (c_int, [c_void_p, c_int32, c_int32, c_uint64, c_uint32, c_int32, c_void_p]),
    "bsc_blr_elbo_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_int32, c_int32, c_double, c_double, c_double,
                                  c_double, c_void_p, c_void_p]),
    "bsc_adam_ascent": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                c_int64, c_double, c_double, c_double, c_double]),
    "bsc_natgrad_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double,
                                   c_double]),
    "bsc_dirichlet_expectation": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "bsc_natgrad_update_f32": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int64, c_float,
                                       c_float]),
    "bsc_softmax_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "bsc_gemm_softmax_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int64,
                                      c_int64, c_int32, c_float, c_void_p, c_int64, c_void_p, c_void_p]),
    "bsc_gemm_softmax_stats": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int64,
                                       c_int64, c_int32, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                       c_void_p]),
    "bsc_suffstats_normal": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "bsc_mog_estep": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p,
                              c_void_p, c_void_p, c_void_p]),
    "bsc_mog_expected_params": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "bsc_mog_natgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_double,
                                c_double]),
    "bsc_mog_expected_params_bound": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p,
                                              c_void_p, c_void_p]),
    "bsc_mog_log_normalizer": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "bsc_mog_natgrad_elbo": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_double,
                                     c_double, c_void_p, c_void_p, c_void_p]),
    "bsc_bbvi_sample": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_uint64, c_uint32,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "bsc_logreg_bbvi_loglik": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                       c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "bsc_bbvi_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                              c_double, c_double, c_double, c_void_p, c_void_p, c_void_p]),
    "bsc_bbvi_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                c_double, c_double, c_double, c_void_p, c_void_p, c_int64, c_double,
                                c_double, c_double, c_double, c_uint64, c_uint32, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "bsc_elemwise": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_int64), c_void_p,
                             POINTER(c_int64), c_int, POINTER(c_void_p), POINTER(c_int64)]),
    "bsc_convert": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_int64), c_void_p,
                            POINTER(c_int64), c_void_p, POINTER(c_int64)]),
    "bsc_sum": (c_int, [c_void_p, c_int, c_int, POINTER(c_int64), POINTER(c_int64), c_int,
                        POINTER(c_int64), POINTER(c_int64), c_void_p, c_void_p]),
    "bsc_lda_sstats": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int32, c_void_p,
                               c_int64, c_void_p, c_int64, c_void_p, c_int64]),
    "bsc_lda_sstats_csc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32,
                                   c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64]),
    "bsc_lda_sstats_bound": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int32, c_void_p,
                                     c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "bsc_lda_sstats_csc_bound": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32,
                                         c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "bsc_dirichlet_expectation_bound": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_double,
                                                c_void_p, c_void_p]),
    "bsc_natgrad_update_f32_elbo": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int64, c_float, c_float,
                                            c_void_p, c_void_p, c_void_p, c_void_p]),
    "bsc_weighted_outer": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                   c_int64, c_int32, c_int32, c_int32, c_double, c_void_p]),
    "bsc_hbm_read_probe": (c_int, [c_void_p, c_void_p, c_size_t, c_int, POINTER(c_double)]),
    "bsc_host_alloc": (c_int, [c_size_t, POINTER(c_void_p)]),
    "bsc_host_free": (c_int, [c_void_p]),
    "bsc_loader_create": (c_int, [c_void_p, c_int64, c_int32, c_int32, POINTER(c_void_p)]),
    "bsc_loader_destroy": (c_int, [c_void_p]),
    "bsc_loader_submit": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64]),
    "bsc_loader_acquire": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p),
                                   POINTER(c_int64)]),
    "bsc_loader_release": (c_int, [c_void_p]),
    "bsc_map_reduce": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_int64), c_int,
                               POINTER(c_int64), c_int, POINTER(c_void_p), POINTER(c_int64),
                               POINTER(c_int64), POINTER(c_int32), POINTER(c_double), c_double,
                               c_double, c_int, c_double, c_void_p, POINTER(c_int64)]),
    "bsc_gemm_strided_batched": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64,
                                         c_void_p, c_int64, c_int64, c_int64,
                                         c_void_p, c_int64, c_int64, c_int64,
                                         c_void_p, c_int64, c_int64, c_int64]),
    "bsc_stream_plan": (c_int, [c_int64, c_int32, c_int64, POINTER(c_int32)]),
    "bsc_gemm_epilogue": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64,
                                  c_void_p, c_int64, c_int64, c_int64,
                                  c_void_p, c_int64, c_int64, c_int64,
                                  c_void_p, c_int64, c_int64, c_int64,
                                  c_int, c_double, c_void_p, c_int64, c_int64, c_int64]),
    "bsc_gemm_fused": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64,
                               c_void_p, c_int64, c_int64, c_int64, c_int,
                               c_void_p, c_int64, c_int64, c_int64, c_int,
                               c_void_p, c_int64, c_int64, c_int64,
                               c_int, c_double, c_void_p, c_int64, c_int64, c_int64, POINTER(c_int32)]),
    "bsc_eye": (c_int, [c_void_p, c_int, c_void_p, c_int64]),
    "bsc_logdet_spd": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_int64,
                               c_int64, c_void_p]),
    "bsc_inverse_spd": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                c_int64, c_void_p, c_void_p]),
}

COMM_ID_BYTES = 128   # BSC_COMM_ID_BYTES
F32, F64 = 0, 1       # bsc_dtype


def dtype_code(dtype):
    """bsc_dtype of a torch / numpy dtype (float32 | float64 only)."""
    name = str(dtype).replace("torch.", "")
    if name == "float32":
        return F32
    if name == "float64":
        return F64
    raise TypeError("bayesic_amd handles float32 and float64, got %s" % (dtype,))


_lib = None


def load_library(path=None):
    """Load (once) and type the shared library.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise BayesicHipError(
            "libbayesic_hip.so not found at %s -- run `python -m bayesic_amd.build` "
            "(there is no CPU fallback)" % path)
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load_library().bsc_last_error()
        raise BayesicHipError("%s failed with status %d: %s" %
                              (what or "libbayesic_hip call", status,
                               msg.decode("utf-8", "replace") if msg else ""))


def ptr(t):
    """Device pointer of a torch tensor (or None / int passthrough)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()
