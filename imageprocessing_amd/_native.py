"""ctypes binding of ``libipamd_hip.so`` (the C ABI declared in ``include/ipamd.h``).

The shared object is built in-tree by ``imageprocessing_amd/csrc/Makefile``.
There is no fallback: if it is missing, or a call fails, an exception is raised.
``torch`` is imported first so that the library binds to the HIP runtime that
PyTorch already loaded (one runtime per process; device pointers and streams
are then shared with torch tensors).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libipamd_hip.so")

_lib = None

_vp = C.c_void_p
_sz = C.c_size_t
_i = C.c_int
_d = C.c_double

# name -> (restype, argtypes); mirrors include/ipamd.h
SIGNATURES = {
    "ipa_version": (_i, []),
    "ipa_last_error": (C.c_char_p, []),
    "ipa_pitch_for": (_sz, [_sz]),
    "ipa_set_option": (_i, [C.c_char_p, _i]),
    "ipa_get_option": (_i, [C.c_char_p, C.POINTER(_i)]),
    "ipa_profile_enable": (_i, [_i]),
    "ipa_profile_stage_count": (_i, []),
    "ipa_profile_stage_name": (C.c_char_p, [_i]),
    "ipa_profile_collect": (_i, [C.POINTER(_d), C.POINTER(_i)]),
    "ipa_bgr2gray_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _vp]),
    "ipa_gray2bgr_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _vp]),
    "ipa_swap_rb_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _vp]),
    "ipa_bitwise_not_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _vp]),
    "ipa_bitwise_or_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _vp]),
    "ipa_fill_rects_c3_u8": (_i, [_vp, _sz, _sz, _i, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "ipa_gray_threshold_u8": (_i, [_vp, _sz, _sz, _vp, _vp, _sz, _sz, _i, _i, _i, _d, _d, _i, _vp]),
    "ipa_threshold_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _d, _d, _i, _vp]),
    "ipa_histogram_u8": (_i, [_vp, _sz, _sz, _i, _i, _i, _vp, _vp]),
    "ipa_threshold_otsu_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _d, _i, _vp, _vp, _vp]),
    "ipa_adaptive_threshold_workspace_bytes": (_sz, [_i, _i, _i]),
    "ipa_adaptive_threshold_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _d, _i, _i, _i, _d, _vp, _sz, _vp]),
    "ipa_gaussian_blur_workspace_bytes": (_sz, [_i, _i, _i]),
    "ipa_gaussian_blur_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _i, _i, _d, _d, _i, _vp, _sz, _vp]),
    "ipa_morph_workspace_bytes": (_sz, [_i, _i, _i]),
    "ipa_morph_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ipa_binary_check_u8": (_i, [_vp, _sz, _sz, _i, _i, _i, _vp, _vp]),
    "ipa_canny_workspace_bytes": (_sz, [_i, _i, _i]),
    "ipa_canny_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _d, _d, _vp, _sz, _vp]),
    "ipa_canny_static_max_bytes": (_sz, []),
    "ipa_canny_static_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _d, _d, _vp]),
    "ipa_hough_dims": (_i, [_i, _i, _d, _d, C.POINTER(_i), C.POINTER(_i)]),
    "ipa_hough_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ipa_hough_segments": (_i, [_i]),
    "ipa_hough_accum_u8": (_i, [_vp, _sz, _sz, _i, _i, _i, _d, _d, _vp, _sz, _vp, _i, _vp, _vp, _sz, _vp]),
    "ipa_hough_peaks_i32": (_i, [_vp, _sz, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "ipa_segments_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ipa_segments_u8": (_i, [_vp, _sz, _sz, _i, _i, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "ipa_hough_lines_p_dims": (_i, [_i, _i, _d, _d, _vp, _vp]),
    "ipa_hough_lines_p_workspace_bytes": (_sz, [_i, _i, _i, _d, _d, _i]),
    "ipa_hough_lines_p_u8": (_i, [_vp, _sz, _sz, _i, _i, _i, _d, _d, _i, _i, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "ipa_front_end_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ipa_front_end_chunks": (_i, [_i, _i, _i]),
    "ipa_front_end_u8": (_i, [_vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _vp, _sz, _vp, _i, _vp,
                              _i, _i, _i, _i, _i, _i, _i, _d, _d, _i, _vp, _sz, _vp]),
    "ipa_median_blur_u8": (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _i, _i, _i, _i, _vp]),
    "ipa_bits_pitch_words": (_sz, [_i]),
    "ipa_front_end_bits_u8": (_i, [_vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _vp, _sz, _vp, _i, _vp,
                                   _i, _i, _i, _i, _i, _i, _i, _d, _d, _i, _vp, _sz, _vp, _vp, _vp]),
    "ipa_segments_bits_u32": (_i, [_vp, _sz, _sz, _i, _i, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
}


class NativeLibraryError(ImportError):
    pass


def lib():
    """The loaded library (loads it on first use; raises if it is not built).  IPA_LIB_PATH names another build
    of the same library (A/B measurements of kernel variants in one session)."""
    global _lib, SO_PATH
    if _lib is None:
        SO_PATH = os.environ.get("IPA_LIB_PATH") or SO_PATH
        if not os.path.exists(SO_PATH):
            raise NativeLibraryError(
                f"{SO_PATH} is missing: build it with `make -C imageprocessing_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
            )
        import torch  # noqa: F401  (loads the HIP runtime this library must share)

        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return lib().ipa_last_error().decode("utf-8", "replace")
