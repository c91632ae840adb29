"""ctypes binding of libtrlda_hip.so (include/trlda_hip.h).

The library is the product: there is no Python or CPU fallback.  If the shared
object is missing, or no GPU is visible, the functions here raise
``RuntimeError`` -- loudly, by design.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# TRLDA_LIB: another build of the SAME library (python -m trlda_amd.build --variant ...: stamps,
# compiler-flag and tuning-constant sweeps under tools/), a development aid -- never a fallback
LIB_PATH = os.environ.get("TRLDA_LIB") or os.path.join(_PKG, "libtrlda_hip.so")

OK = 0
ERR_ARG, ERR_SHAPE, ERR_WORD_ID, ERR_NO_DEVICE, ERR_HIP, ERR_VALUE = -1, -2, -3, -4, -5, -6
SSTATS_SEGMENTED, SSTATS_ATOMIC = 0, 1

i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="F_CONTIGUOUS")
vp = C.c_void_p

_SIGNATURES = {
    # name: (restype, argtypes)
    "trlda_last_error": (C.c_char_p, []),
    "trlda_version": (C.c_int, []),
    "trlda_device_count": (C.c_int, []),
    "trlda_seed": (None, [C.c_uint]),
    "trlda_rng_get_state": (None, [np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")]),
    "trlda_rng_set_state": (None, [np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")]),
    "trlda_sample_gamma": (None, [C.c_int, C.c_int, C.c_int, f64p]),
    "trlda_sample_gamma_init": (None, [C.c_int, C.c_int, f64p]),
    "trlda_estep": (C.c_int, [C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, f64p, f64p, f64p, f64p,
                              C.c_int, C.c_double, vp, C.c_int]),
    "trlda_mstep_blend": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, f64p,
                                    f64p, f64p, C.c_int]),
    "trlda_tr_init": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, i32p,
                                i32p, i32p, f64p, f64p, C.c_int]),
    "trlda_docs_from_text": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "trlda_docs_from_buffer": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(vp)]),
    "trlda_docs_num_docs": (C.c_int64, [vp]),
    "trlda_docs_nnz": (C.c_int64, [vp]),
    "trlda_docs_offsets": (C.POINTER(C.c_int64), [vp]),
    "trlda_docs_ids": (C.POINTER(C.c_int32), [vp]),
    "trlda_docs_cnts": (C.POINTER(C.c_int32), [vp]),
    "trlda_docs_destroy": (C.c_int, [vp]),
    "trlda_dev_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(vp)]),
    "trlda_dev_free": (C.c_int, [C.c_int, vp]),
    "trlda_dev_upload": (C.c_int, [C.c_int, vp, vp, C.c_size_t]),
    "trlda_dev_download": (C.c_int, [C.c_int, vp, vp, C.c_size_t]),
    "trlda_dev_synchronize": (C.c_int, [C.c_int]),
    "trlda_batch_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, i32p, i32p, i32p]),
    "trlda_batch_destroy": (C.c_int, [vp]),
    "trlda_batch_num_docs": (C.c_int, [vp]),
    "trlda_batch_nnz": (C.c_int64, [vp]),
    "trlda_batch_max_doc_len": (C.c_int, [vp]),
    "trlda_batch_long_word_len": (C.c_int, [vp]),
    "trlda_batch_num_long_words": (C.c_int, [vp]),
    "trlda_model_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int]),
    "trlda_model_destroy": (C.c_int, [vp]),
    "trlda_model_set_stream": (C.c_int, [vp, vp]),
    "trlda_model_set_sstats_mode": (C.c_int, [vp, C.c_int]),
    "trlda_model_set_dense_preamble": (C.c_int, [vp, C.c_int]),
    "trlda_model_set_doc_threads": (C.c_int, [vp, C.c_int]),
    "trlda_model_synchronize": (C.c_int, [vp]),
    "trlda_model_set_split_docs": (C.c_int, [vp, C.c_int]),
    "trlda_model_set_draw_ahead": (C.c_int, [vp, C.c_int]),
    "trlda_model_lane_state": (C.c_int, [vp]),
    "trlda_model_lane_timing": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "trlda_model_inlaunch_draws": (C.c_longlong, [vp]),
    "trlda_model_set_aux_decay": (C.c_int, [vp, C.c_int]),
    "trlda_model_inlaunch_decays": (C.c_longlong, [vp]),
    "trlda_model_dp_direct_alloc": (C.c_int, [vp, C.c_size_t, C.c_int, vp]),
    "trlda_model_dp_direct_connect": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "trlda_model_dp_direct_close": (C.c_int, [vp]),
    "trlda_model_last_split_workgroups": (C.c_int, [vp]),
    "trlda_model_set_allgatherv": (C.c_int, [vp, vp, vp]),
    "trlda_model_set_word_sharding": (C.c_int, [vp, C.c_int]),
    "trlda_model_last_word_sharded": (C.c_int, [vp]),
    "trlda_model_set_merged_launch": (C.c_int, [vp, C.c_int]),
    "trlda_model_set_split_lists": (C.c_int, [vp, C.c_int]),
    "trlda_batch_num_very_long_words": (C.c_int, [vp]),
    "trlda_model_last_merged": (C.c_int, [vp]),
    "trlda_model_set_lambda": (C.c_int, [vp, f64p]),
    "trlda_model_get_lambda": (C.c_int, [vp, f64p]),
    "trlda_model_set_alpha": (C.c_int, [vp, f64p]),
    "trlda_model_lambda_dev": (vp, [vp]),
    "trlda_model_get_sstats": (C.c_int, [vp, f64p]),
    "trlda_model_estep": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_double, vp]),
    "trlda_model_estep_io": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_double, vp]),
    "trlda_model_estep_io_next": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_double, vp]),
    "trlda_model_set_prefetch": (C.c_int, [vp, C.c_int]),
    "trlda_model_set_deferred_stats": (C.c_int, [vp, C.c_int]),
    "trlda_model_flush": (C.c_int, [vp]),
    "trlda_model_last_deferred": (C.c_int, [vp]),
    "trlda_model_set_stream_lanes": (C.c_int, [vp, C.c_int]),
    "trlda_model_estep_io_ahead": (C.c_int, [vp, vp, C.POINTER(C.c_void_p), C.c_int, vp, vp, vp, C.c_int,
                                             C.c_double, vp]),
    "trlda_model_estep_corpus": (C.c_int, [vp, C.c_int64, vp, vp, vp, C.c_int, vp, vp, C.POINTER(C.c_void_p),
                                           C.c_int, C.c_int, C.c_double, vp]),
    "trlda_model_lane_steps": (C.c_longlong, [vp]),
    "trlda_model_get_lane_timing": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "trlda_model_estep_host": (C.c_int, [vp, vp, f64p, f64p, C.c_int, C.c_double, vp]),
    "trlda_model_blend": (C.c_int, [vp, vp, vp, C.c_double, C.c_double, C.c_double]),
    "trlda_model_tr_init": (C.c_int, [vp, vp, vp, C.c_double, C.c_double, C.c_int]),
    "trlda_model_wordcounts": (C.c_int, [vp, vp, vp]),
    "trlda_model_tr_init_wc": (C.c_int, [vp, vp, vp, C.c_double, C.c_double, C.c_double]),
    "trlda_model_copy_lambda": (C.c_int, [vp, vp]),
    "trlda_model_online_update": (C.c_int, [vp, vp, C.c_int, C.c_double, C.c_int, C.c_int,
                                            C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                            C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                            vp]),
    "trlda_model_batch_update": (C.c_int, [vp, vp, C.c_double, C.c_int, C.c_int, C.c_int,
                                           C.c_double, vp]),
    "trlda_model_cumulative_update": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double,
                                                vp]),
    "trlda_model_lower_bound": (C.c_int, [vp, vp, f64p, C.c_double, C.c_double, C.c_int, C.c_double,
                                C.POINTER(C.c_double)]),
    "trlda_model_allreduce_sstats": (C.c_int, [vp, vp, vp]),
    "trlda_model_online_update_multi": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double,
                                                  C.c_int, C.c_int, C.c_double, C.c_double,
                                                  C.c_double, C.c_int, C.c_double,
                                                  C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "trlda_model_set_allgather": (C.c_int, [vp, vp, vp]),
    "trlda_model_online_update_dp": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int,
                                               C.POINTER(C.c_int32), C.c_int, C.c_double, C.c_int,
                                               C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                               C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "trlda_model_estep_dp": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(C.c_int32), vp, vp,
                                       vp, C.c_int, C.c_double, vp, C.c_int, vp, C.c_double,
                                       C.c_double, C.c_double]),
    "trlda_model_set_fused_update": (C.c_int, [vp, C.c_int]),
    "trlda_model_set_next_preamble": (C.c_int, [vp, C.c_int]),
    "trlda_model_set_carry_rowsums": (C.c_int, [vp, C.c_int]),
    "trlda_model_set_keep_sstats": (C.c_int, [vp, C.c_int]),
    "trlda_model_d2h_bytes": (C.c_int64, [vp]),
    "trlda_model_set_host_gamma_draw": (C.c_int, [vp, C.c_int]),
    "trlda_model_sample_gamma": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_double, vp]),
    "trlda_model_sample_gamma_cols": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_double, vp]),
    "trlda_model_estep_resident": (C.c_int, [vp, vp, C.c_int, C.c_double]),
    "trlda_model_eb_gamma_stats": (C.c_int, [vp, C.c_int, vp, f64p]),
    "trlda_model_eb_gamma_stats_multi": (C.c_int, [vp, vp, C.c_int, vp, f64p]),
    "trlda_model_batch_update_multi": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_int,
                                                 C.c_int, C.c_int, C.c_double]),
    "trlda_model_batch_update_dp": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                              C.c_double, C.c_int, C.c_int, C.c_int, C.c_double]),
    "trlda_model_allreduce": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "trlda_model_estep_resident_shard": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double]),
    "trlda_model_eb_lambda_stats": (C.c_int, [vp, C.POINTER(C.c_double), f64p]),
    "trlda_eb_online_alpha_step": (C.c_int, [C.c_int, f64p, f64p, C.c_double, C.c_double, C.c_double,
                                             f64p]),
    "trlda_eb_online_eta_step": (C.c_double, [C.c_double, C.c_double, f64p, C.c_int, C.c_int, C.c_double,
                                              C.c_double]),
    "trlda_eb_alpha_line_search": (C.c_int, [C.c_int, f64p, f64p, C.c_double, C.c_int, C.c_double,
                                             C.c_double, f64p]),
    "trlda_eb_set_verbosity": (None, [C.c_int]),
    "trlda_eb_eta_line_search": (C.c_double, [C.c_double, C.c_double, f64p, C.c_int, C.c_int, C.c_int,
                                              C.c_double, C.c_double]),
    "trlda_model_online_eb": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                        C.c_double, C.c_double, f64p, C.POINTER(C.c_double)]),
    "trlda_model_online_eb_begin": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "trlda_model_online_eb_finish": (C.c_int, [vp, C.c_double, C.c_double, C.c_double, f64p,
                                               C.POINTER(C.c_double)]),
    "trlda_model_online_eb_pending": (C.c_int, [vp]),
    "trlda_debug_host_psi": (None, [C.c_int, f64p, f64p, f64p]),
    "trlda_model_adaptive_stats": (C.c_int, [vp, C.c_double, C.c_double, C.c_double,
                                            C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "trlda_model_adaptive_stats_dev": (C.c_int, [vp, vp, vp, C.c_double, C.c_double, C.c_double,
                                                C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "trlda_debug_fold16": (C.c_int, [C.c_int, vp, vp, vp, vp]),
    "trlda_debug_peek": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
    "trlda_debug_merged_stamps": (C.c_int, [vp, vp]),
    "trlda_model_set_doc_kernel": (C.c_int, [vp, C.c_int]),
    "trlda_model_last_doc_kernel": (C.c_char_p, [vp]),
    "trlda_model_set_split_preamble": (C.c_int, [vp, C.c_int]),
    "trlda_model_last_preamble_fused": (C.c_int, [vp]),
    "trlda_debug_digamma": (C.c_int, [C.c_int, C.c_int, C.c_double, vp, vp, vp, vp, vp]),
    "trlda_model_set_timing": (C.c_int, [vp, C.c_int]),
    "trlda_model_get_timing": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double),
                                         C.POINTER(C.c_int64)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own HIP / HSA runtime (torch/lib/libamdhip64.so, soname
    libamdhip64.so.7 -- the soname libtrlda_hip.so asks for).  A process must hold ONE runtime: if
    /opt/rocm's copy is loaded first, torch later loads its own beside it and finds no GPU.  So
    when a torch with a bundled runtime is installed, that copy is loaded first (without
    importing torch) and libtrlda_hip.so binds to it -- the same result as `import torch` before
    `import trlda_amd`.  TRLDA_NO_TORCH_RUNTIME=1 skips this."""
    if os.environ.get("TRLDA_NO_TORCH_RUNTIME") == "1":
        return
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if not spec or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    """Load (once) and return the ctypes handle; raises if the .so was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "trlda_amd: %s not found -- build it with `python -m trlda_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


class TrldaError(RuntimeError):
    """A TRLDA::Exception-equivalent (reference include/exception.h -> RuntimeError)."""

    def __init__(self, code, message):
        RuntimeError.__init__(self, message)
        self.code = code


def check(rc):
    if rc != OK:
        raise TrldaError(rc, lib().trlda_last_error().decode() or "trlda_hip error %d" % rc)


def device_count():
    return lib().trlda_device_count()


def require_gpu():
    if device_count() < 1:
        raise TrldaError(ERR_NO_DEVICE, "trlda_amd needs an AMD GPU (gfx950); no HIP device is "
                                        "visible and there is no CPU fallback")
