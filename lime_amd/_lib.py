"""ctypes loader of liblime_hip.so (the C ABI in include/lime_hip.h).

The library is the product: if it is missing or cannot be loaded this module raises -- there
is no Python or CPU substitute for the kernels.
"""
from __future__ import annotations

import ctypes as C
import os

_DIR = os.path.dirname(os.path.abspath(__file__))
# LIME_LIB: another build of the library (ablation / timing variants of tools/*.sh) -- loaded INSTEAD of the installed one, which those scripts
# used to overwrite; bench.py refuses to run with it
LIB_PATH = os.environ.get("LIME_LIB") or os.path.join(_DIR, "liblime_hip.so")

LIME_OK = 0
ERR_ARG, ERR_HIP, ERR_NOMEM, ERR_MAXLEN, ERR_HALO, ERR_DOCID, ERR_IO = -1, -2, -3, -4, -5, -6, -7
MAX_CLUSTER = 65536
TILE = 4096


class Cluster(C.Structure):
    _fields_ = [("pStart", C.c_uint64), ("len", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("n_clusters", C.c_uint64), ("max_len", C.c_uint64), ("n_updates", C.c_uint64),
                ("n_cross", C.c_uint32), ("n_big", C.c_uint32), ("flags", C.c_uint32), ("wave_records_max", C.c_uint32), ("edge", C.c_uint32), ("reserved", C.c_uint32)]


class Records(C.Structure):
    _fields_ = [("n_bins", C.c_uint32), ("bin_shift", C.c_uint32), ("d_recs", C.c_void_p), ("d_binbase", C.c_void_p),
                ("d_bigrecs", C.c_void_p), ("n_bigrecs", C.c_uint64)]


class LimeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lime error {code}: {msg}")
        self.code = code


# every symbol include/lime_hip.h declares: (restype, argtypes)
_vp, _u32, _u64, _i, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_size_t
_pp = C.POINTER(C.c_void_p)
_pu64 = C.POINTER(C.c_uint64)
SYMBOLS = {
    "lime_init": (_i, [_i, _pp]),
    "lime_shutdown": (None, [_vp]),
    "lime_last_error": (C.c_char_p, []),
    "lime_free": (None, [_vp]),
    "lime_version": (C.c_char_p, []),
    "lime_device_count": (_i, []),
    "lime_pick_device": (_i, [C.c_uint]),
    "lime_set_option": (_i, [_vp, C.c_char_p, C.c_char_p]),
    "lime_trim_cache": (_sz, []),
    "lime_reserve": (C.c_int, [_sz]),
    "lime_detect": (_i, [_vp, _vp, _vp, _u64, _u32, _u32, _pp, _pu64, _pu64]),
    "lime_detect_to_file": (_i, [_vp, _vp, _vp, _u64, _u32, _u32, C.c_char_p, _pu64, _pu64]),
    "lime_register_file": (_i, [_vp, _sz, _i]),
    "lime_unregister_file": (None, [_vp]),
    "lime_score": (_i, [_vp, _vp, _vp, _u64, _vp, _u64, _u32, _u32, _vp]),
    "lime_fused": (_i, [_vp, _vp, _vp, _vp, _u64, _u32, _u32, _u32, _vp, _pu64, _pu64]),
    "lime_choose": (_i, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "lime_sim_bytes": (_sz, [_u32, _u32]),
    "lime_fused_dev": (_i, [_vp, _vp, _vp, _vp, _u64, _u64, _i, _u32, _u32, _u32, _vp, _i, _vp]),
    "lime_detect_dev": (_i, [_vp, _vp, _vp, _u64, _u64, _i, _u64, _u32, _u32, _pp, _pu64, _pu64, _vp]),
    "lime_score_dev": (_i, [_vp, _vp, _vp, _u64, _vp, _u64, _u32, _u32, _vp, _i, _vp]),
    "lime_choose_dev": (_i, [_vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "lime_synth_dev": (_i, [_vp, _u64, _u64, _u64, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "lime_get_stats": (_i, [_vp, C.POINTER(Stats), _vp]),
    "lime_set_timing": (_i, [_vp, _i]),
    "lime_get_timing": (_i, [_vp, C.POINTER(C.c_double), _pu64]),
    "lime_get_timing_ex": (_i, [_vp, C.POINTER(C.c_double), _pu64]),
    "lime_get_host_times": (_i, [_vp, C.POINTER(C.c_double)]),
    "lime_sym_index": (C.c_uint8, [C.c_uint8]),
    "lime_pair_score": (C.c_uint8, [_vp, _vp]),
    "lime_write_clrs": (_i, [C.c_char_p, _vp, _u64]),
    "lime_write_aux": (_i, [C.c_char_p, _u32, _u32, _u32, _u64, _u64]),
    "lime_read_aux": (_i, [C.c_char_p, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32), _pu64, _pu64]),
    "lime_write_res_txt": (_i, [C.c_char_p, _vp, _vp, _u32, _u32, _u32, C.c_float]),
    "lime_write_res_bin": (_i, [C.c_char_p, C.c_char_p, _vp, _vp, _u32, _u32, _u32, C.c_float]),
    "lime_classify": (_i, [_u32, _vp, _i, _u32, _u32, C.c_char_p, C.c_char_p, _i, _i, _vp]),
    "lime_classify_error": (C.c_char_p, []),
    "lime_fused_stream": (_i, [_vp, _vp, _vp, _vp, _u64, _u32, _u32, _u32, _u64, _vp, _pu64, _pu64]),
    "lime_choose_pairs_dev": (_i, [_vp, _vp, _u32, _u32, _u32, C.c_float, _vp, _vp, _pp, _pu64, _vp]),
    "lime_fused_choose_dev": (_i, [_vp, _vp, _vp, _vp, _u64, _u32, _u32, _u32, _u32, C.c_float, _vp, _vp, _pp, _pu64, C.POINTER(Stats), _vp]),
    "lime_score_choose": (_i, [_vp, _vp, _vp, _u64, _vp, _u64, _u32, _u32, _u32, C.c_float, _vp, _vp, _pp, _pu64, _vp]),
    "lime_combine_edges": (_i, [_vp, _u32]),
    "lime_comm_unique_id": (_i, [_vp]),
    "lime_comm_init": (_i, [_vp, _i, _i, _pp]),
    "lime_comm_destroy": (None, [_vp]),
    "lime_comm_count": (_i, [_vp, C.POINTER(_i)]),
    "lime_comm_error": (C.c_char_p, []),
    "lime_comm_reduce_scatter_tables": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "lime_comm_allreduce_tables": (_i, [_vp, _vp, _sz, _vp]),
    "lime_comm_combine_counters": (_i, [_vp, _vp, _vp]),
    "lime_records_layout": (_i, [_vp, _u32, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "lime_fused_records_dev": (_i, [_vp, _vp, _vp, _vp, _u64, _u64, _i, _u32, _u32, _u32, _vp]),
    "lime_records_get": (_i, [_vp, _vp, _vp, _vp]),
    "lime_apply_records_dev": (_i, [_vp, _u32, _vp, _vp, _u32, _u32, _vp, _u64, _u64, _u64, _vp, _vp]),
    "lime_comm_exchange_records": (_i, [_vp, _vp, _u32, _u32, _vp, _sz, _pu64, _pu64, _vp]),
    "lime_fused_multi": (_i, [_i, _vp, _vp, _vp, _vp, _u64, _u32, _u32, _u32, _vp, _pu64, _pu64]),
    "lime_score_choose_multi": (_i, [_i, _vp, _vp, _vp, _u64, _vp, _u64, _u32, _u32, _u32, C.c_float, _vp, _vp, _pp, _pu64]),
    "lime_write_res_txt_pairs": (_i, [C.c_char_p, _vp, _vp, _vp, _u32, _u32, C.c_float]),
    "lime_write_res_bin_pairs": (_i, [C.c_char_p, C.c_char_p, _vp, _vp, _vp, _u32, _u32, C.c_float]),
}

_LIB = None


def load():
    """Load liblime_hip.so (importing torch first so both share one HIP runtime)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C lime_amd/csrc` "
                          "(or __graft_entry__.build()); lime_amd has no fallback path")
    try:
        import torch  # noqa: F401  (loads libamdhip64.so.7 that the library binds to)
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc):
    if rc != LIME_OK:
        raise LimeError(rc, load().lime_last_error().decode(errors="replace"))


def hip_memcpy_d2d(dst, src, nbytes):
    """device-to-device copy through the HIP runtime torch already loaded (tests: library-owned lists)"""
    hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return hip.hipMemcpy(dst, src, nbytes, 3)          # hipMemcpyDeviceToDevice
