"""ctypes binding of libspurfies_hip.so (the C ABI declared in include/spurfies_hip.h).

The product path fails loudly when the HIP library is missing: there is no CPU or PyTorch
fallback for any kernel.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPF_LIB_PATH") or os.path.join(HERE, "lib", "libspurfies_hip.so")      # SPF_LIB_PATH: same-box A/B of two builds (tools/ab_prof.sh)


class GridConfig(C.Structure):
    _fields_ = [("voxel_size", C.c_float * 3), ("voxel_scale", C.c_int32 * 3), ("kernel_size", C.c_int32 * 3),
                ("max_points_per_voxel", C.c_int32), ("max_occ_voxels", C.c_int32), ("ranges", C.c_float * 6), ("compat", C.c_int32)]


class GridInfo(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("cell", C.c_float * 3), ("dims", C.c_int32 * 3),
                ("n_points", C.c_int32), ("n_in_range", C.c_int32), ("n_occupied", C.c_int32), ("max_cell_points", C.c_int32)]


_P = C.c_void_p
_I = C.c_int32
_F = C.c_float

# name -> (restype, argtypes): exactly the entry points of include/spurfies_hip.h
class LossWeights(C.Structure):
    """spf_loss_weights"""
    _fields_ = [("rgb", C.c_float), ("eikonal", C.c_float), ("tv", C.c_float), ("local", C.c_float), ("pseudo", C.c_float),
                ("world", C.c_int32)]


class WgradProblem(C.Structure):
    """spf_wgrad_problem"""
    _fields_ = [("G", C.c_void_p), ("A", C.c_void_p), ("lda", C.c_int32), ("dW", C.c_void_p), ("ldw", C.c_int32), ("dbias", C.c_void_p),
                ("C", C.c_int32), ("layout", C.c_int32), ("col_rot", C.c_int32), ("col_mod", C.c_int32), ("n_rows", C.c_void_p), ("max_rows", C.c_int32)]


class ProloguePacks(C.Structure):
    """spf_prologue_packs"""
    _fields_ = ([(n, C.c_void_p) for n in ("cw0", "cb0", "cw2", "cb2", "cw4", "cb4", "c_packed", "c_zero")] + [("c_zero_floats", C.c_int64)] +
                [(n, C.c_void_p) for n in ("rw6", "rb6", "rw0", "rb0", "rw2", "rb2", "rw4", "rb4", "r_packed", "r_zero")] + [("r_zero_floats", C.c_int64)])


class LocalTermsArgs(C.Structure):
    """spf_local_terms"""
    _fields_ = [("lsum", C.c_void_p), ("lfirst", C.c_void_p), ("desc", C.c_void_p), ("lscale", C.c_void_p)]


LOCAL_MAX_VIEWS = 8


class LocalDesc(C.Structure):
    """spf_local_desc (built on the host, uploaded as bytes)"""
    _fields_ = [("feat", C.c_uint64 * LOCAL_MAX_VIEWS), ("cam", C.c_float * (LOCAL_MAX_VIEWS * 32)), ("center", C.c_float * 3), ("size", C.c_float),
                ("n_src", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32)]


class SamplerEvalArgs(C.Structure):
    """spf_sampler_eval_args"""
    _fields_ = [("z", C.c_void_p), ("n", C.c_int32), ("n_prev", C.c_int32), ("sdf_prev", C.c_void_p), ("merged_idx", C.c_void_p), ("pair_tmp", C.c_void_p),
                ("pair_off", C.c_void_p), ("slot_point", C.c_void_p), ("sdf_cur", C.c_void_p), ("beta", C.c_void_p), ("beta0", C.c_void_p),
                ("eps", C.c_float), ("bound_coef", C.c_float), ("add_tiny", C.c_float), ("beta_iters", C.c_int32), ("u_more", C.c_void_p), ("N_more", C.c_int32),
                ("u_fin", C.c_void_p), ("N_fin", C.c_int32), ("z_merged", C.c_void_p), ("merged_out", C.c_void_p), ("points_new", C.c_void_p),
                ("sel", C.c_void_p), ("Ne", C.c_int32), ("near", C.c_float), ("far", C.c_float), ("cam_loc", C.c_void_p), ("ray_dirs", C.c_void_p),
                ("z_out", C.c_void_p), ("points_out", C.c_void_p), ("SR", C.c_int32), ("slot_sample", C.c_void_p), ("ray_valid", C.c_void_p),
                ("flags", C.c_void_p), ("it", C.c_int32)]


SIGNATURES = {
    "spf_abi_version": (C.c_int, []),
    "spf_last_error": (C.c_char_p, []),
    "spf_grid_create": (C.c_int, [C.POINTER(GridConfig), C.POINTER(_P)]),
    "spf_grid_destroy": (None, [_P]),
    "spf_grid_build": (C.c_int, [_P, _P, _I, _P]),
    "spf_grid_get_info": (C.c_int, [_P, C.POINTER(GridInfo)]),
    "spf_grid_query": (C.c_int, [_P, _P, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P]),
    "spf_grid_knn": (C.c_int, [_P, _P, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P]),
    "spf_grid_sweep_hits": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, C.c_int64, C.c_int64, _P, _F, _P, _P, _P, _P]),
    "spf_compact_points": (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P]),
    "spf_compact_sync_words": (C.c_int64, [C.c_int64]),
    "spf_compact_pairs": (C.c_int, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P]),
    "spf_compact_pairs_filter": (C.c_int, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "spf_voxel_cells": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_float * 3), _F, _P, _P]),
    "spf_geo_packed_floats": (C.c_int64, []),
    "spf_geo_pack": (C.c_int, [_P] * 14),
    "spf_build_pairs": (C.c_int, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "spf_geo_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _P]),
    "spf_geo_clock_read": (C.c_int, [_P, _I]),
    "spf_geo_backward_latents": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P]),
    "spf_color_packed_floats": (C.c_int64, []),
    "spf_color_pack": (C.c_int, [_P] * 8 + [C.c_int64, _P]),
    "spf_color_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "spf_color_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "spf_rhead_packed_floats": (C.c_int64, []),
    "spf_rhead_pack": (C.c_int, [_P] * 10 + [C.c_int64, _P]),
    "spf_rhead_forward": (C.c_int, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "spf_rhead_backward": (C.c_int, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "spf_sampler_uniform": (C.c_int, [_P, _P, _P, _P, _I, _I, _F, _F, _P, _P, _P]),
    "spf_sampler_iter": (C.c_int, [_P, _P, _P, _P, _I, _I, _F, _F, _I, _I, _F, _P, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "spf_sampler_eval": (C.c_int, [C.POINTER(SamplerEvalArgs), _P, _I, _I, _P]),
    "spf_sampler_finish": (C.c_int, [_P, _I, _P, _I, _P, _I, _F, _F, _P, _P, _I, _P, _P, _P, _I, _P]),
    "spf_sampler_train": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _P, _I, _P, _I, _F, _F, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "spf_filter_points": (C.c_int, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "spf_render_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P]),
    "spf_render_rgb": (C.c_int, [_P, _P, _I, _I, _P, _P]),
    "spf_render_rgb_backward": (C.c_int, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "spf_render_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "spf_local_forward": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "spf_local_backward": (C.c_int, [_P, _P, _P, _I, _I, _P, _P]),
    "spf_wgrad_workspace_floats": (C.c_int64, [_I]),
    "spf_wgrad": (C.c_int, [_P, _P, _I, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "spf_wgrad_batched": (C.c_int, [C.POINTER(WgradProblem), _I, _P, _I, _P, _I, _I, _P]),
    "spf_scatter_add_rows": (C.c_int, [_P, _P, C.c_int64, _I, _P, _P]),
    "spf_tv_forward": (C.c_int, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "spf_tv_backward": (C.c_int, [_P, _P, _P, _P, _P, _I, _F, _I, _I, _P, _P, _P]),
    "spf_fixed_accumulate": (C.c_int, [_P, _P, C.c_int64, _P]),
    "spf_camera_rays": (C.c_int, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _F, _P, _P]),
    "spf_camera_uniform": (C.c_int, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P, _I, _F, _F, _P, _P, _P, _P, _P, _P, _I, _I, _P, C.POINTER(ProloguePacks), _P]),
    "spf_adam_workspace_floats": (C.c_int64, []),
    "spf_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _I, _P, _P, _P]),
    "spf_loss_workspace_floats": (C.c_int64, []),
    "spf_loss_forward": (C.c_int, [_P, _P, _P, _P, _I, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _I, _P, _I, C.POINTER(LossWeights), _P, _P, _P, _P,
                                   C.POINTER(LocalTermsArgs), _P]),
    "spf_loss_backward_finalize": (C.c_int, [_P, C.POINTER(LossWeights), _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, C.c_int64, _P, _P, _P, _P, _P, _P,
                                             _P, _P, _P, _P, _I, _P, C.POINTER(LocalTermsArgs), _P]),
    "spf_loss_backward": (C.c_int, [_P, _P, C.POINTER(LossWeights), _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, C.POINTER(LocalTermsArgs), _P]),
}

_lib = None


class SpurfiesHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SpurfiesHipError(
                f"{LIB_PATH} is missing: build it with `python -m spurfies_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no fallback path.")
        import torch  # noqa: F401  (loads torch's libamdhip64 first so both share one HIP runtime)

        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(code: int, what: str = ""):
    if code != 0:
        msg = lib().spf_last_error()
        raise SpurfiesHipError(f"{what} failed ({code}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
