"""ctypes binding of libmimo_hip.so (the C-ABI declared in include/mimo_hip.h).

This is the ONLY compute path of the package: there is no PyTorch/CPU fallback.  If the
shared library is missing the import of any op raises; if an entry point rejects its
arguments a `MimoHipError` is raised with the C error code.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIMO_HIP_LIB selects another build of the same sources (tools/microbench.py: libmimo_hip_tune.so, whose tuning knobs
# are read from the environment); the default is the shipped library.
LIB_PATH = os.environ.get("MIMO_HIP_LIB") or os.path.join(_HERE, "libmimo_hip.so")

F16, BF16, F32 = 0, 1, 2
EPI_SILU, EPI_GEGLU, EPI_OUT_F32, EPI_RES_F32, EPI_NO_SPLITK = 1, 2, 4, 8, 16

c_vp, c_i, c_i64, c_f, c_u = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint
c_sz = ctypes.c_size_t


class ConvParams(ctypes.Structure):
    _fields_ = [(n, c_i) for n in ("n", "Hin", "Win", "Cin", "Hout", "Wout", "Cout", "ksize", "stride",
                                   "pad_t", "pad_l", "Hup", "Wup", "Cin2", "imgs_per_bias_row",
                                   "img_bias_ld")]


class EpilogueExt(ctypes.Structure):  # mimo_epilogue_ext
    _fields_ = [("colstats", c_vp), ("ln_out", c_vp), ("ln_gamma", c_vp), ("ln_beta", c_vp), ("ln_pe", c_vp),
                ("ln_eps", c_f), ("ln_pe_frames", c_i), ("ln_rows_per_frame", c_i64),
                ("row_half", c_vp), ("row_stats", c_vp), ("a_row_stats", c_vp), ("a_colsum", c_vp), ("a_slots", c_i), ("a_eps", c_f)]


class HconvParams(ctypes.Structure):  # mimo_hconv_params
    _fields_ = [(n, c_i) for n in ("n", "H", "W", "Cout", "upsample2x", "imgs_per_bias_row", "img_bias_ld")]


class CompositeParams(ctypes.Structure):  # mimo_composite_params
    _fields_ = [("crop", c_vp), ("mask", c_vp), ("bk", c_vp), ("occ", c_vp), ("vid", c_vp), ("prev", c_vp), ("out", c_vp),
                ("factor", ctypes.c_double)] + [(n, c_i) for n in ("pad_h", "pad_w", "top", "bottom", "left", "right",
                                                                 "w_min", "h_min", "mh", "mw", "H", "W")]


# name -> argtypes, mirrors include/mimo_hip.h one to one
SIGNATURES = {
    "mimo_version": [],
    "mimo_reload_tuning": [],
    "mimo_workspace_bytes": [],
    "mimo_row_stat_slots": [c_i],
    "mimo_gemm": [c_i, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i, c_i, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_f, c_u, c_vp,
                  c_sz, c_vp],
    "mimo_gemm_ext": [c_i, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i, c_i, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_f, c_u,
                      c_vp, c_sz, ctypes.POINTER(EpilogueExt), c_vp],
    "mimo_conv2d": [c_i, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(ConvParams), c_vp, c_vp, c_vp, c_f, c_u, c_vp, c_sz, c_vp],
    "mimo_conv2d_ext": [c_i, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(ConvParams), c_vp, c_vp, c_vp, c_f, c_u, c_vp, c_sz,
                        ctypes.POINTER(EpilogueExt), c_vp],
    "mimo_conv3x3_tapsum": [c_vp, c_i64, c_i, c_i, c_i, c_i, c_vp, c_vp, c_f, c_vp],
    "mimo_group_norm_stats_cols": [c_vp, c_i, c_vp, c_i, c_i, c_i64, c_i, c_f, c_vp, c_vp],
    "mimo_group_norm_stats_slabs": [c_vp, c_i, c_i, c_vp, c_i, c_i, c_i, c_i64, c_i, c_f, c_vp, c_vp],
    "mimo_group_norm_stats": [c_vp, c_i, c_vp, c_i, c_i, c_i, c_i, c_i64, c_i, c_f, c_vp, c_vp, c_i, c_vp],
    "mimo_group_norm_apply": [c_vp, c_i, c_vp, c_i, c_i, c_i, c_i, c_i64, c_i, c_vp, c_vp, c_vp, c_i, c_vp, c_vp, c_vp],
    "mimo_group_norm_apply_split3": [c_vp, c_i, c_i, c_i, c_i64, c_i, c_vp, c_vp, c_vp, c_i, c_vp, c_i64, c_i, c_i, c_vp],
    "mimo_group_norm_affine": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_vp],
    "mimo_conv3x3_fused": [c_i, c_vp, c_i, c_vp, c_i, c_vp, c_i, c_vp, c_i64, c_vp, ctypes.POINTER(HconvParams), c_vp, c_vp, c_vp,
                           c_vp, c_vp, c_f, c_u, c_vp],
    "mimo_layer_norm": [c_vp, c_i, c_i, c_i64, c_i, c_f, c_vp, c_vp, c_vp, c_i64, c_i, c_vp, c_vp, c_vp],
    "mimo_attention": [c_i, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                       c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_vp],
    "mimo_attention_fp8qk": [c_i, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                       c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_vp],
    "mimo_ff_fused": [c_i, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i, c_vp],
    "mimo_ff_proj_fused": [c_i, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i, c_vp,
                           c_vp],
    "mimo_block_tail_fused": [c_i, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_f, c_vp, c_vp, c_vp, c_vp,
                              c_vp, c_i64, c_vp, c_i64, c_i64, c_i, c_vp, c_vp],
    "mimo_block_head_fused": [c_i, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_f, c_vp, c_i64, c_i,
                              c_vp, c_i64, c_vp, c_i64, c_i64, c_i, c_vp],
    "mimo_temporal_attention": [c_i, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_i64, c_i, c_i, c_f, c_vp],
    "mimo_softmax_rows": [c_i, c_vp, c_i64, c_vp, c_i64, c_i64, c_i, c_f, c_vp],
    "mimo_ncfhw_to_tokens": [c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_i, c_i, c_i64, c_i, c_vp, c_vp],
    "mimo_tokens_to_ncfhw": [c_vp, c_i, c_i, c_i64, c_i, c_i, c_i, c_i, c_i, c_f, c_vp, c_vp],
    "mimo_cast": [c_vp, c_i, c_i, c_i64, c_vp, c_vp],
    "mimo_cfg_ddim_step": [c_vp, c_vp, c_vp, c_i, c_i, c_i64, c_i, c_f, c_f, c_f, c_f, c_f, c_vp],
    "mimo_cfg_ddim_step_frames": [c_vp, c_vp, c_vp, c_i, c_i, c_i64, c_vp, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_vp],
    "mimo_frames_differ": [c_vp, c_i, c_i64, c_vp, c_vp],
    "mimo_window_accumulate": [c_vp, c_i64, c_vp, c_i, c_i, c_i, c_i, c_i64, c_vp, c_vp, c_vp],
    "mimo_tokens_to_image": [c_vp, c_i, c_i, c_i64, c_i, c_i, c_i, c_vp, c_vp],
    "mimo_resample_pass_u8": [c_vp, c_i, c_i64, c_i64, c_i64, c_i64, c_vp, c_i, c_i, c_i, c_i, c_vp, c_vp, c_i, c_i, c_vp],
    "mimo_u8_to_tokens": [c_i, c_vp, c_i64, c_i, c_i, c_i, c_vp, c_vp],
    "mimo_u8_to_planar_f32": [c_vp, c_i, c_i64, c_i, c_f, c_vp, c_vp, c_vp, c_vp],
    "mimo_composite_frame": [ctypes.POINTER(CompositeParams), c_vp],
}


class MimoHipError(RuntimeError):
    pass


_lib = None


RESTYPES = {"mimo_workspace_bytes": c_sz}  # every other entry point returns int (0 or an error code)


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.  If this library were loaded first it would pull in
    # the system runtime, torch would then allocate on one runtime and these kernels launch on the other (first launch fails
    # with hipErrorNoDevice).  Importing torch first makes the loader resolve libamdhip64 to the copy torch already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise MimoHipError(
            f"{LIB_PATH} is missing: build it with `python -m mimo_amd.build` "
            "(hipcc, gfx950). mimo_amd has no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, c_i)
    _lib = lib
    return lib


def call_int(name, *args):
    """Entry points that return a value instead of a status (mimo_row_stat_slots, mimo_version)."""
    return int(getattr(load(), name)(*args))


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise MimoHipError(f"{name} failed with code {rc}"
                           + (" (MIMO_EINVAL)" if rc == -1 else " (MIMO_EDTYPE)" if rc == -2 else
                              f" (hipError_t {rc})" if rc > 0 else ""))
    return rc
