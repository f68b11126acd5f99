"""ctypes binding of libnlc_hip.so (the C ABI declared in include/nlc_hip.h).

The library is built in-tree by ``csrc/build.sh`` (hipcc --offload-arch=gfx950).  There is no
CPU fallback: if the shared object is missing or a symbol is absent, importing the ops fails
loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("NLC_HIP_LIB", _HERE / "libnlc_hip.so"))     # override: kernel A/B experiments only
BUILD_SCRIPT = _HERE / "csrc" / "build.sh"

ABI_VERSION = 7                 # NLC_ABI_VERSION of include/nlc_hip.h
NLC_F32, NLC_BF16, NLC_F16 = 0, 1, 2
MATH_NATIVE, MATH_F16X3 = 0, 1      # nlc_conv_desc.math / nlc_pack_conv_weights_ex
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
OUT_NHWC, OUT_NCHW_F32 = 0, 1
SCHED_VARIANTS = {
    "ddim": 0, "ddim_simple": 1, "ddim_simple_orig": 2, "ddim_simple_drag": 3,
    "ddpm": 4, "ddpm_orig": 5, "ddim_orig": 6,
}
CLIP_MODES = {"none": 0, "clamp": 1, "dynamic": 2}
VAR_MODES = {"none": 0, "fixedsmall": 1, "fixedlarge": 2, "learned": 3}


class NlcError(RuntimeError):
    """Raised for every failure of the HIP path.  ``rc`` is the library's return code when one is known (NLC_EINVAL -1: the
    call was rejected before anything was launched; NLC_ELAUNCH -2: a launch failed; NLC_EUNSUPPORTED -3)."""

    def __init__(self, msg="", rc=None):
        super().__init__(msg)
        self.rc = rc


NLC_EINVAL, NLC_ELAUNCH, NLC_EUNSUPPORTED = -1, -2, -3


class GnIn(C.Structure):
    """nlc_gn_in: GroupNorm (+FiLM) (+SiLU) of a convolution's input, from the ride-along totals of its producers."""
    _fields_ = [
        ("stats0", C.c_void_p), ("stats1", C.c_void_p),
        ("granule0", C.c_int32), ("granule1", C.c_int32),
        ("groups", C.c_int32),
        ("eps", C.c_float),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("ss_stride", C.c_int32),
        ("act", C.c_int32),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("x0", C.c_void_p), ("x1", C.c_void_p),
        ("C0", C.c_int32), ("C1", C.c_int32),
        ("B", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
        ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("upsample2x", C.c_int32),
        ("w", C.c_void_p),
        ("Cin_pad", C.c_int32), ("Cout_pad", C.c_int32),
        ("bias", C.c_void_p),
        ("emb", C.c_void_p),
        ("emb_stride", C.c_int32),
        ("res", C.c_void_p),
        ("out_scale", C.c_float),
        ("act", C.c_int32),
        ("out", C.c_void_p),
        ("out_mode", C.c_int32),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64),
        ("stats_out", C.c_void_p),
        ("stats_bytes", C.c_int64),
        ("policy", C.c_int32),
        ("gn_coef", C.c_void_p),
        ("gn_act", C.c_int32),
        ("tuning", C.c_int32),
        ("res_upsample2x", C.c_int32),
        ("stats_granule", C.c_int32),
        ("math", C.c_int32),
        ("debug", C.c_int32),
        ("w_scale", C.c_void_p),
        ("norm_out", C.c_void_p),
        ("gn_in", C.POINTER(GnIn)),
    ]


class SchedDesc(C.Structure):
    _fields_ = [
        ("xt", C.c_void_p), ("eps_out", C.c_void_p), ("noise", C.c_void_p),
        ("known", C.c_void_p), ("mask", C.c_void_p),
        ("sigma_t", C.c_void_p), ("sigma_prev", C.c_void_p),
        ("eps_norm_sumsq", C.c_void_p),
        ("dyn_s", C.c_void_p),
        ("logvar_ext", C.c_void_p),
        ("x0", C.c_void_p), ("x_prev", C.c_void_p), ("eps_used", C.c_void_p),
        ("B", C.c_int32), ("C", C.c_int32), ("Cnet", C.c_int32), ("HW", C.c_int32),
        ("variant", C.c_int32), ("clip", C.c_int32), ("var_mode", C.c_int32),
        ("phases", C.c_int32),
        ("eta", C.c_float), ("min_var_coef", C.c_float),
    ]


class SigmaDesc(C.Structure):
    _fields_ = [
        ("sumsq", C.c_void_p), ("sigma_in", C.c_void_p), ("t_in", C.c_void_p), ("sigmas", C.c_void_p), ("t_slopes", C.c_void_p),
        ("sigma_t", C.c_void_p), ("sigma_prev", C.c_void_p), ("t", C.c_void_p), ("c_in", C.c_void_p),
        ("sqrt_dim", C.c_float), ("norm_max", C.c_float), ("norm_min", C.c_float),
        ("sigma_sched", C.c_float), ("sigma_prev_sched", C.c_float), ("t_sched", C.c_float), ("time_shift", C.c_float),
        ("refine", C.c_int32), ("prev_is_ratio", C.c_int32), ("n_sigmas", C.c_int32), ("B", C.c_int32),
    ]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol include/nlc_hip.h declares
SIGNATURES = {
    "nlc_version": (C.c_int, []),
    "nlc_last_error": (C.c_char_p, []),
    "nlc_conv_pack_dims": (C.c_int, [_i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nlc_pack_conv_weights": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "nlc_pack_conv_weights_ex": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "nlc_conv2d": (C.c_int, [C.POINTER(ConvDesc), _i, _vp]),
    "nlc_conv2d_workspace_bytes": (C.c_int64, [C.POINTER(ConvDesc), _i]),
    "nlc_conv2d_stats_partials": (C.c_int, [C.POINTER(ConvDesc), _i]),
    "nlc_conv2d_prologue_supported": (C.c_int, [C.POINTER(ConvDesc), _i]),
    "nlc_conv2d_norm_out_supported": (C.c_int, [C.POINTER(ConvDesc), _i]),
    "nlc_conv2d_gn_in_supported": (C.c_int, [C.POINTER(ConvDesc), _i]),
    "nlc_resblock_small": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _vp, _i, _vp]),
    "nlc_groupnorm_coef": (C.c_int, [_i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "nlc_conv_first": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "nlc_conv_first_stats_partials": (C.c_int, [_i, _i, _i, _i, _i, _i, _i]),
    "nlc_groupnorm_workspace_bytes": (C.c_int64, [_i, _i, _i, _i]),
    "nlc_groupnorm": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "nlc_groupnorm_prestats": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp]),
    "nlc_groupnorm_pool2x2": (C.c_int, [_vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "nlc_attention": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "nlc_avgpool2x2": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "nlc_upsample2x": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "nlc_pad_rb": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "nlc_nhwc_to_nchw_f32": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "nlc_nchw_f32_to_nhwc": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "nlc_timestep_embedding": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "nlc_row_sumsq": (C.c_int, [_vp, _vp, _i, _i64, _i64, _vp]),
    "nlc_refine_sigma": (C.c_int, [_vp, _f, _f, _f, _f, _f, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "nlc_refine_sigma_ex": (C.c_int, [C.POINTER(SigmaDesc), _vp]),
    "nlc_sigma_correct": (C.c_int, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "nlc_proj_sigma": (C.c_int, [_vp, _f, _f, _f, _f, _f, _f, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "nlc_dynamic_threshold": (C.c_int, [_vp, _f, _f, _vp, _i, _i64, _vp]),
    "nlc_dynamic_threshold_ws_bytes": (C.c_int64, [_i]),
    "nlc_dynamic_threshold_ws": (C.c_int, [_vp, _f, _f, _vp, _i, _i64, _vp, _i64, _vp]),
    "nlc_sched_x0": (C.c_int, [C.POINTER(SchedDesc), _vp]),
    "nlc_sched_step": (C.c_int, [C.POINTER(SchedDesc), _vp, _vp]),
    "nlc_scale_rows": (C.c_int, [_vp, _vp, _f, _vp, _i, _i64, _vp]),
    "nlc_lincomb_rows": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "nlc_cast_f64_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "nlc_row_sumsq_f64": (C.c_int, [_vp, _vp, _i, _i64, _vp]),
    "nlc_row_cosine_f64": (C.c_int, [_vp, _vp, C.c_double, _vp, _i, _i64, _vp]),
    "nlc_edm_scalars": (C.c_int, [_vp, _f, _vp, _vp, _vp, _vp, _i, _vp]),
    "nlc_edm_eps": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "nlc_f64_lincomb": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "nlc_inpaint_A": (C.c_int, [_vp, _vp, _vp, _i, _i, _i64, _i64, _vp]),
    "nlc_inpaint_Apinv": (C.c_int, [_vp, _vp, _vp, _i, _i, _i64, _i64, _vp]),
}

_lib = None


def build(force: bool = False) -> Path:
    """Compile the HIP sources for gfx950 (works without a GPU)."""
    if force:
        for o in (_HERE / "csrc" / "obj").glob("*.o"):
            o.unlink()
    subprocess.run(["bash", str(BUILD_SCRIPT)], check=True)
    return LIB_PATH


def load() -> C.CDLL:
    """Load libnlc_hip.so and bind every declared symbol.  Raises NlcError when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise NlcError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            f"Build it with `bash {BUILD_SCRIPT}` or `python -c 'import __graft_entry__ as g; g.build()'`."
        )
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise NlcError(f"libnlc_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    ver = lib.nlc_version()
    if ver != ABI_VERSION:
        raise NlcError(f"libnlc_hip.so ABI version {ver} != {ABI_VERSION}: rebuild it with `bash {BUILD_SCRIPT}`")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().nlc_last_error().decode("utf-8", "replace")
        raise NlcError(f"{what} failed (rc={rc}): {msg}", rc=rc)


def pack_dims(dtype: int):
    a, b = C.c_int(), C.c_int()
    check(load().nlc_conv_pack_dims(dtype, C.byref(a), C.byref(b)), "nlc_conv_pack_dims")
    return a.value, b.value
