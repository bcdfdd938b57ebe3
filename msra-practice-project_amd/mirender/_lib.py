"""ctypes binding of libmirender.so (the C ABI declared in include/mi_render.h).

The HIP library is the product: there is no CPU or PyTorch fallback.  If the shared object
is missing or a symbol is absent, importing/using this module raises immediately.
torch must be imported first so the process has ONE HIP runtime (torch's bundled
libamdhip64.so.7 satisfies libmirender's NEEDED entry by SONAME).
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (loads libamdhip64 before libmirender)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmirender.so")

c_f32p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_f32 = ctypes.c_float
_f64 = ctypes.c_double
_u64 = ctypes.c_uint64
_vp = ctypes.c_void_p

# name -> (restype, argtypes): every symbol include/mi_render.h declares
SIGNATURES = {
    "mi_abi_version": (_int, []),
    "mi_last_error": (ctypes.c_char_p, []),
    "mi_field_num_params": (_int, [_int]),
    "mi_field_packed_floats": (_i64, [_int]),
    "mi_field_macs": (_i64, [_int]),
    "mi_field_param_shape": (_int, [_int, _int, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "mi_field_pack": (_int, [_int, ctypes.POINTER(_vp), _int, _f32, _vp, _vp]),
    "mi_field_eval_points": (_int, [_int, _vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "mi_field_eval_rays": (_int, [_int, _vp, _vp, _vp, _vp, _i64, _i64, _int, _vp, _vp]),
    "mi_gen_rays": (_int, [_int, _int, _f64, ctypes.POINTER(_f32), _i64, _i64, _vp, _int, _vp]),
    "mi_sample_coarse": (_int, [_i64, _f32, _f32, _int, _vp, _vp, _u64, _u64, _vp, _vp]),
    "mi_composite": (_int, [_i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi_sample_fine": (_int, [_i64, _f32, _f32, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi_sample_fine_pos": (_int, [_i64, _f32, _f32, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi_merge_raw": (_int, [_i64, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "mi_split_grad": (_int, [_i64, _int, _int, _vp, _vp, _vp, _int, _vp, _vp]),
    "mi_sample_pdf": (_int, [_i64, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "mi_image_metrics_workspace_floats": (_i64, [_int, _int, _int, _int]),
    "mi_image_metrics": (_int, [_vp, _vp, _int, _int, _int, _int, _vp, _int, _vp, _vp, _vp]),
    "mi_grid_points": (_int, [_int, _vp, _f32, _i64, _i64, _vp, _vp]),
    "mi_nerf_loss_workspace_floats": (_i64, [_i64]),
    "mi_nerf_loss": (_int, [_i64, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi_ray_bank": (_int, [_int, _int, _f64, _vp, _vp, _int, _i64, _vp, _int, _vp]),
    "mi_adam_step": (_int, [_int, ctypes.POINTER(_int), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                            ctypes.POINTER(_vp), ctypes.POINTER(_i64), _f32, _f32, _f32, _f32, _f32, _f32,
                            ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]),
    "mi_render_workspace_bytes": (_i64, [_i64, _int, _int]),
    "mi_render_shared_field_extra_bytes": (_i64, [_i64, _int, _int]),
    "mi_render_rays": (_int, [_int, _vp, _int, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _int, _int, _vp, _vp, _vp,
                              _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mi_composite_bwd": (_int, [_i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi_field_packed_bwd_floats": (_i64, [_int]),
    "mi_field_pack_bwd": (_int, [_int, ctypes.POINTER(_vp), _int, _f32, _vp, _vp]),
    "mi_field_train_acts_floats": (_i64, [_int]),
    "mi_field_train_grads_floats": (_i64, [_int]),
    "mi_field_bwd_partial_floats": (_i64, [_i64]),
    "mi_field_eval_rays_train": (_int, [_int, _vp, _vp, _vp, _vp, _i64, _i64, _int, _vp, _vp, _vp]),
    "mi_field_eval_points_train": (_int, [_int, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "mi_field_film_partial_floats": (_i64, [_i64, _i64]),
    "mi_field_backward": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, ctypes.POINTER(_vp),
                                 ctypes.POINTER(_vp), _int, _vp, _vp]),
    "mi_event_create": (_vp, []),
    "mi_event_destroy": (None, [_vp]),
    "mi_event_record": (_int, [_vp, _vp]),
    "mi_event_elapsed_ms": (_int, [_vp, _vp, ctypes.POINTER(_f32)]),
    "mi_render_set_mlp_events": (None, [_vp, _vp, _vp, _vp]),
}

ABI_VERSION = 4      # include/mi_render.h as of this binding (mi_abi_version() of the library must equal it)

_lib = None


class MiRenderError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MiRenderError(
            f"{LIB_PATH} not found: build it with `python msra-practice-project_amd/csrc/build.py` "
            "(the HIP library is required; there is no fallback path)")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    # the signatures below are those of ONE ABI version: a stale or foreign library (MI_DIAG_LIB in tests/conftest.py makes
    # loading another one a supported path) must be refused before any call passes it arguments in the wrong order
    try:
        lib.mi_abi_version.restype = ctypes.c_int
        have = lib.mi_abi_version()
    except AttributeError:
        have = None
    if have != ABI_VERSION:
        raise MiRenderError(f"{LIB_PATH} has ABI version {have}, this binding is written against {ABI_VERSION}: rebuild it "
                            "with `python msra-practice-project_amd/csrc/build.py --force`")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().mi_last_error().decode(errors="replace")
        raise MiRenderError(f"{what} failed ({rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
