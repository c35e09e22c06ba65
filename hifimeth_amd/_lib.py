"""ctypes loader of libhifimeth_hip.so (the C ABI of include/hifimeth_hip.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded this module raises,
and hm_create() fails when no gfx950 device is present.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HM_LIB_PATH") or os.path.join(HERE, "libhifimeth_hip.so")  # HM_LIB_PATH: diagnostic builds
CSRC = os.path.join(HERE, "csrc")
WEIGHTS_DIR = os.path.join(HERE, "weights")


class hm_call_t(C.Structure):
    _fields_ = [("read_id", C.c_int32), ("qoff", C.c_int32), ("strand", C.c_uint8), ("ctx", C.c_uint8),
                ("scaled_prob", C.c_uint8), ("reserved", C.c_uint8), ("p", C.c_float)]


class hm_read_t(C.Structure):
    _fields_ = [("read_id", C.c_int32), ("l_qseq", C.c_int32), ("flag", C.c_int32), ("width", C.c_uint8 * 4),
                ("seq4", C.c_void_p), ("kin", C.c_void_p * 4)]


class hm_timing_t(C.Structure):
    _fields_ = [("prep_ms", C.c_double), ("scan_ms", C.c_double), ("emit_ms", C.c_double), ("window_ms", C.c_double),
                ("front_ms", C.c_double * 3), ("tail_ms", C.c_double * 3),
                ("prep_launches", C.c_int64), ("scan_launches", C.c_int64), ("emit_launches", C.c_int64),
                ("window_launches", C.c_int64), ("front_launches", C.c_int64 * 3), ("tail_launches", C.c_int64 * 3),
                ("front_sites", C.c_int64 * 3), ("window_sites", C.c_int64),
                ("pack_ms", C.c_double), ("empty_ms", C.c_double),
                ("pack_launches", C.c_int64), ("empty_launches", C.c_int64),
                ("trunk_ms", C.c_double * 3), ("edge_ms", C.c_double * 3),
                ("trunk_launches", C.c_int64 * 3), ("edge_launches", C.c_int64 * 3),
                ("trunk_positions", C.c_int64 * 3),
                ("trunk_list_steps", C.c_int64 * 3), ("trunk_const_steps", C.c_int64 * 3),
                ("group_bases", C.c_int64), ("group_bytes", C.c_int64), ("tail_strip_passes", C.c_int64)]


HM_ABI_VERSION = 5   # include/hifimeth_hip.h


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hifimeth_amd has no CPU fallback)")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  If this library were loaded
    # first, a later `import torch` would bring a SECOND HIP runtime into the process and torch would then see no
    # GPU; importing torch first makes the loader resolve our NEEDED libamdhip64.so.7 to the copy already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, cp = C.c_void_p, C.c_int32, C.c_int64, C.c_char_p
    sig = {
        "hm_create": (C.c_int, [C.POINTER(vp), cp, C.c_int, C.c_int]),
        "hm_destroy": (None, [vp]),
        "hm_last_error": (cp, [vp]),
        "hm_set_option": (C.c_int, [vp, cp, i64]),
        "hm_submit_read": (C.c_int, [vp, i32, i32, i32, vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int]),
        "hm_clear": (C.c_int, [vp]),
        "hm_upload": (C.c_int, [vp]),
        "hm_run": (C.c_int, [vp]),
        "hm_sync": (C.c_int, [vp]),
        "hm_num_sites": (i64, [vp, C.c_int]),
        "hm_fetch": (i64, [vp, vp, i64]),
        "hm_flush": (C.c_int, [vp]),
        "hm_drain": (i64, [vp, vp, i64]),
        "hm_batch_begin": (vp, [vp]),
        "hm_batch_submit_read": (C.c_int, [vp, i32, i32, i32, vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int]),
        "hm_batch_submit_reads": (i64, [vp, vp, i64, C.c_int, vp]),
        "hm_trunk_mask_for_reads": (C.c_int, [vp, i64, C.c_int]),
        "hm_batch_staged_bases": (i64, [vp]),
        "hm_batch_enqueue": (C.c_int, [vp]),
        "hm_batch_done": (C.c_int, [vp]),
        "hm_batch_wait": (i64, [vp, C.POINTER(vp)]),
        "hm_batch_num_sites": (i64, [vp, C.c_int]),
        "hm_batch_release": (C.c_int, [vp]),
        "hm_scan_sites": (i64, [vp, C.c_int, vp, vp, vp, i64]),
        "hm_windows": (C.c_int, [vp, C.c_int, i64, i64, vp]),
        "hm_cnn_logits": (C.c_int, [vp, C.c_int, vp, i64, vp, vp, vp]),
        "hm_debug_layer": (i64, [vp, C.c_int, vp, C.c_int, vp, i64]),
        "hm_convert_model": (C.c_int, [cp, cp]),
        "hm_get_stamps": (C.c_int, [vp, C.POINTER(C.c_uint64), C.c_int]),
        "hm_get_timing": (C.c_int, [vp, C.POINTER(hm_timing_t)]),
        "hm_abi_version": (C.c_int, []),
        "hm_timing_size": (C.c_size_t, []),
        "hm_reset_timing": (C.c_int, [vp]),
        # pileup
        "hm_pileup_create": (C.c_int, [C.POINTER(vp), C.c_int]),
        "hm_pileup_destroy": (None, [vp]),
        "hm_pileup_last_error": (cp, [vp]),
        "hm_pileup_set_option": (C.c_int, [vp, cp, C.c_double]),
        "hm_pileup_set_reference": (C.c_int, [vp, i32, vp, vp]),
        "hm_pileup_use_planes": (C.c_int, [vp, vp, vp, vp]),
        "hm_pileup_planes": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]),
        "hm_pileup_submit_read": (C.c_int, [vp, C.c_uint32, i32, i32, i64, i32, i32, vp, i32, vp, i64, vp]),
        "hm_pileup_run": (C.c_int, [vp]),
        "hm_pileup_num_records": (i64, [vp]),
        "hm_pileup_histograms": (C.c_int, [vp, vp]),
        "hm_pileup_fetch_records": (i64, [vp, vp, vp, vp, vp, i64]),
        "hm_pileup_label_histograms": (C.c_int, [vp, vp, i64, vp]),
        "hm_pileup_count": (C.c_int, [vp, vp]),
        "hm_pileup_fetch_loci": (i64, [vp, vp, vp, vp, i64, i64, i64, vp, i64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._hm_symbols = tuple(sig)
    # a library built from another header would write past (or short of) the structs mirrored above
    if L.hm_abi_version() != HM_ABI_VERSION or L.hm_timing_size() != C.sizeof(hm_timing_t):
        raise ImportError(f"{LIB_PATH}: ABI {L.hm_abi_version()} / hm_timing_t {L.hm_timing_size()} bytes, this package expects "
                          f"{HM_ABI_VERSION} / {C.sizeof(hm_timing_t)}: rebuild (python -c 'import __graft_entry__ as g; g.build()')")
    _lib = L
    return L
