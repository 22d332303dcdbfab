"""ctypes binding of libgsseg_hip.so (the C ABI declared in include/gsseg.h).

The library is built in-tree by `build()` (hipcc --offload-arch=gfx950; cross-compiles without a
GPU) and loaded from this package directory.  There is NO fallback: if the shared object is missing
or fails to load, every op raises.  torch is used only for device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSSEG_LIB", os.path.join(_HERE, "libgsseg_hip.so"))
CSRC = os.path.join(_HERE, "csrc")

GS_F16, GS_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY02, ACT_TANH = 0, 1, 2, 3
GS_MAX_TAPS = 64
ABI_VERSION = 50


class GsConvGeom(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in (
        "N", "IH", "IW", "Cin", "in_pix_stride", "in_coff", "OHg", "OWg", "Cout", "OH", "OW",
        "out_pix_stride", "out_coff", "isy", "isx", "osy", "osx", "ooy", "oox", "ntaps")] + [
        ("tap_dy", c_int32 * GS_MAX_TAPS), ("tap_dx", c_int32 * GS_MAX_TAPS), ("tap_w", c_int32 * GS_MAX_TAPS)] + [
        (n, c_int32) for n in ("Dg", "Din", "Dout", "isz", "osz", "ooz")] + [("tap_dz", c_int32 * GS_MAX_TAPS)]


# name -> (restype, argtypes): exactly the declarations of include/gsseg.h
class GsPackDesc(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("w_fwd", c_void_p), ("w_dgrad", c_void_p)] + [
        (n, c_int32) for n in ("Cout", "Cin", "taps", "transposed")]


GS_SEG_MAX = 4


class GsSegPackDesc(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("pack", c_void_p)] + [(n, c_int32) for n in ("Cout", "Cin", "taps", "transposed", "nseg")] + [
        ("kind", c_int32 * GS_SEG_MAX), ("ci0", c_int32 * GS_SEG_MAX), ("len", c_int32 * GS_SEG_MAX)]


class GsQ8PackDesc(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("pack", c_void_p), ("wexp", c_void_p)] + [(n, c_int32) for n in ("Cout", "Cin", "taps")]


_P, _F = c_void_p, c_void_p   # device pointers are passed as integers
PROTOTYPES = {
    "gs_last_error": (c_char_p, []),
    "gs_abi_version": (c_int, []),
    "gs_conv_igemm_mtiles": (c_int, [POINTER(GsConvGeom)]),
    "gs_conv3x3_stat_rows": (c_int, [c_int] * 6),
    "gs_upconv2x2_wgrad_parts": (c_int, [c_int] * 5),
    "gs_upconv2x2_wgrad_ws_floats": (c_int64, [c_int] * 5),
    "gs_upconv2x2_wgrad_slabs": (c_int, [_P, _P, _F] + [c_int] * 14 + [c_void_p]),
    "gs_set_persistent_grid": (c_int, [c_int]),
    "gs_get_persistent_grid": (c_int, []),
    "gs_conv_igemm": (c_int, [POINTER(GsConvGeom), _P, _P, _P, _F, _F, c_int, c_int, _F, c_int64, c_void_p]),
    "gs_conv_igemm_workspace_floats": (c_int64, []),
    "gs_conv_igemm_batch": (c_int, [c_int, POINTER(POINTER(GsConvGeom)), _P, POINTER(c_void_p), _P, _F, POINTER(c_void_p),
                                    c_int, c_int, _F, c_int64, c_void_p]),
    "gs_upconv2x2_fwd": (c_int, [_P, _P, _F, _P] + [c_int] * 18 + [c_void_p]),
    "gs_upconv2x2_dgrad": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 14 + [c_void_p]),
    "gs_conv3x3_mtiles": (c_int, [c_int, c_int, c_int, c_int]),
    "gs_conv3x3_set_kernel_form": (c_int, [c_int]),
    "gs_conv3d_3x3x3_mtiles": (c_int, [c_int] * 5),
    "gs_conv3d_3x3x3": (c_int, [_P, _P, _P, _F, _F] + [c_int] * 10 + [POINTER(c_int32)] * 3 + [c_int, c_int, c_void_p]),
    "gs_conv3d_3x3x3_wgrad_ws_floats": (c_int64, [c_int] * 6),
    "gs_conv3d_3x3x3_wgrad_parts": (c_int, [c_int] * 6),
    "gs_conv3d_3x3x3_wgrad_slabs": (c_int, [_P, _P, _F] + [c_int] * 11 + [c_void_p]),
    "gs_conv3d_3x3x3_wgrad": (c_int, [_P, _P, _F] + [c_int] * 11 + [c_void_p]),
    "gs_upsample2x_bilinear_fwd": (c_int, [_P, _P] + [c_int] * 13 + [c_void_p]),
    "gs_upsample2x_bilinear_bwd": (c_int, [_P, _P] + [c_int] * 13 + [c_void_p]),
    "gs_affine_warp": (c_int, [_F, _F, _F, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "gs_optim_chunk_elems": (c_int, []),
    "gs_optim_rmsprop": (c_int, [_P] * 7 + [c_int] + [c_float] * 6 + [c_void_p]),
    "gs_optim_adam": (c_int, [_P] * 7 + [c_int, _P] + [c_float] * 5 + [c_void_p]),
    "gs_conv3x3": (c_int, [_P, _P, _P, _F, _F] + [c_int] * 9 + [POINTER(c_int32), POINTER(c_int32), c_int, c_int, c_void_p]),
    "gs_conv3x3_wgrad": (c_int, [_P, _P, _F] + [c_int] * 10 + [c_void_p]),
    "gs_conv3x3_wgrad_ws_floats": (c_int64, [c_int] * 5),
    "gs_conv3x3_wgrad_parts": (c_int, [c_int] * 5),
    "gs_conv3x3_wgrad_slabs": (c_int, [_P, _P, _F] + [c_int] * 10 + [c_void_p]),
    "gs_wgrad_reduce_unpack": (c_int, [_F, c_int, _F, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "gs_conv_wgrad": (c_int, [POINTER(GsConvGeom), _P, _P, _F, c_int, c_void_p]),
    "gs_conv_wgrad_assign": (c_int, [POINTER(GsConvGeom), _P, _P, _F, c_int, c_void_p]),
    "gs_conv_wgrad_single_pass": (c_int, [POINTER(GsConvGeom)]),
    "gs_conv_wgrad_parts": (c_int, [POINTER(GsConvGeom)]),
    "gs_conv_wgrad_ws_floats": (c_int64, [POINTER(GsConvGeom)]),
    "gs_conv_wgrad_slabs": (c_int, [POINTER(GsConvGeom), _P, _P, _F, c_int, c_void_p]),
    "gs_conv_wgrad_slabs_batch": (c_int, [c_int, POINTER(POINTER(GsConvGeom)), _P, _P, _F, c_int, c_void_p]),
    "gs_conv_smallcin_mtiles": (c_int, [c_int, c_int, c_int]),
    "gs_conv_smallcin_fwd": (c_int, [_F, _F, _F, _P, _F] + [c_int] * 12 + [c_void_p]),
    "gs_conv_direct_wgrad_ws_floats": (c_int64, [c_int] * 6),
    "gs_conv_smallcin_wgrad": (c_int, [_F, _P, _F, _F] + [c_int] * 10 + [c_float, c_int, c_void_p]),
    "gs_conv_smallcin_dgrad": (c_int, [_P, _F, _F] + [c_int] * 10 + [c_float, c_int, c_void_p]),
    "gs_conv_smallcout_fwd": (c_int, [_P, _F, _F, _F] + [c_int] * 11 + [c_void_p]),
    "gs_conv_smallcout_bwd": (c_int, [_P, _F, _F, _P, _F, _F, _F] + [c_int] * 10 + [c_float, c_int, c_void_p]),
    "gs_bn_partials_floats": (c_int64, [c_int, c_int]),
    "gs_bn_finalize": (c_int, [_F, c_int, c_int, c_double, _F, _F, _F, _F, c_float, c_float, _F, _F, _F, _F, c_void_p]),
    "gs_bn_eval_coeffs": (c_int, [c_int, _F, _F, _F, _F, c_float, _F, _F, _F, _F, c_void_p]),
    "gs_bn_act_apply": (c_int, [_P, _F, _F, c_int, _P, c_int, c_int, _P, _P, c_float] + [c_int] * 5 + [c_void_p]),
    "gs_bn_bwd_tiles": (c_int, [c_int, c_int, c_int]),
    "gs_bn_bwd_tiles_used": (c_int, [c_int, c_int, c_int, c_int]),
    "gs_bn_act_bwd_reduce": (c_int, [_P, _P, c_int, c_int, _P, _P, c_int, _P, c_float, _F, _F, _F, _F, c_int, _F]
                             + [c_int] * 5 + [c_void_p]),
    "gs_bn_bwd_coeffs": (c_int, [_F, c_int, c_int, c_double, c_float, _F, _F, _F, _F, c_void_p]),
    "gs_bn_act_bwd_apply": (c_int, [_P, _P, c_int, c_int, _P, _P, c_int, _P, c_float, _F, _F, _F, _F, _F, _F, c_int,
                                    c_int, _P]
                            + [c_int] * 5 + [c_void_p]),
    "gs_maxpool3d_fwd": (c_int, [_P, c_int, c_int, _P] + [c_int] * 6 + [c_void_p]),
    "gs_maxpool2x2_fwd": (c_int, [_P, c_int, c_int, _P] + [c_int] * 5 + [c_void_p]),
    "gs_maxpool2x2_fwd_pair": (c_int, [_P, _P, c_int, _P, _P, c_int] + [c_int] * 5 + [c_void_p]),
    "gs_maxpool3d_bwd": (c_int, [_P, c_int, c_int, _P, _P, c_int, c_int, _P] + [c_int] * 6 + [c_void_p]),
    "gs_colsum": (c_int, [_P, c_int, c_int] + [c_int] * 8 + [c_float, _F, _F, c_int, c_void_p]),
    "gs_stem_bn_bwd_wgrad": (c_int, [_P, _P, c_int, c_int, _F, _F, _F, _F, _F, _F, _F, c_int, _F, _F, c_int, c_int, c_int,
                                     c_float, c_int, c_void_p]),
    "gs_stem_stats": (c_int, [_F, _F, _F, _F, c_int, c_int, c_int, c_void_p]),
    "gs_stem_bwd_onepass": (c_int, [_F, _P, _P, c_int, c_int, c_int, _F, _F, c_int, c_int, c_int, c_int, c_void_p]),
    "gs_stem_bwd_finalize": (c_int, [_F, _F, _F, _F, _F, _F, _F, c_int, c_float, _F, _F, _F, c_int, c_int, c_int, c_void_p]),
    "gs_stem_fwd_bn": (c_int, [_F, _F, _F, _F, c_int, _P, c_int, c_int, c_int, c_int, c_void_p]),
    "gs_stem_bwd_tiles": (c_int, [c_int, c_int, c_int]),
    "gs_head1x1_bn_fwd": (c_int, [_P, _F, _F, c_int, _F, _F, _F] + [c_int] * 5 + [c_void_p]),
    "gs_head1x1_bn_wgrad": (c_int, [_P, _F, _F, c_int, _F, _F, _F, _F, _F] + [c_int] * 4 + [c_float, c_int, c_void_p]),
    "gs_bn_act_bwd_reduce_head": (c_int, [_P, _F, _F, c_int, _F, _F, _F, _F, c_int, _F] + [c_int] * 5 + [c_void_p]),
    "gs_bn_act_bwd_apply_head": (c_int, [_P, _F, _F, c_int, _F, _F, _F, _F, _F, _F, c_int, _P] + [c_int] * 5 + [c_void_p]),
    "gs_bn_partials_colsum": (c_int, [_F, c_int, c_int, c_int, c_int, c_float, _F, c_void_p]),
    "gs_pack_weight": (c_int, [_F, _P, _P, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gs_pack_weight_multi": (c_int, [c_int, POINTER(GsPackDesc), c_int, c_void_p]),
    "gs_unpack_wgrad": (c_int, [_F, _F, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "gs_upconv_merge_pack": (c_int, [_F, _F, _F, _F, _P, _P, _F, c_int, c_int, c_int, c_void_p]),
    "gs_fake_postprocess_ws_floats": (c_int64, [c_int]),
    "gs_isic_fake_trans_ws_bytes": (c_int64, [c_int, c_int64]),
    "gs_isic_fake_trans": (c_int, [_F, _F, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int,
                                   c_float, c_float, c_void_p]),
    "gs_fake_postprocess": (c_int, [_F, _F, _F, _F, c_int, c_int64, c_void_p]),
    "gs_upconv8_image_fwd": (c_int, [_P, c_int, c_int, _P, c_int, _F, _F, _P] + [c_int] * 7 + [c_void_p]),
    "gs_upconv8_image_wgrad_ok": (c_int, [c_int, c_int]),
    "gs_upconv8_image_wgrad_ws_floats": (c_int64, [c_int] * 4),
    "gs_upconv8_image_wgrad": (c_int, [_P, c_int, _P, c_int, _F, _F] + [c_int] * 5 + [c_void_p]),
    "gs_upconv_split_wgrad": (c_int, [_F, _F, _F, _F, _F, c_float, _F, _F, _F, _F, c_int, c_int, c_void_p]),
    "gs_upconv_split_wgrad_det": (c_int, [_F, _F, _F, _F, _F, c_float, _F, _F, _F, _F, _F, c_int, c_int, c_void_p]),
    "gs_upconv_split_wgrad_ws_floats": (c_int64, [c_int, c_int]),
    "gs_upconv_split_wgrad_parts_ok": (c_int, [c_int, c_int]),
    "gs_upconv_split_wgrad_parts": (c_int, [_F, c_int, c_int64, _F, _F, _F, _F, c_float, _F, _F, _F, _F, _F, c_int, c_int, c_void_p]),
    "gs_nchw_to_nhwc": (c_int, [_F, _P] + [c_int] * 7 + [c_void_p]),
    "gs_nhwc_to_nchw": (c_int, [_P, c_int, c_int, _F] + [c_int] * 4 + [c_float, c_int, c_void_p]),
    "gs_seg_loss_fwd": (c_int, [_F, _P, c_int, c_int, c_int, c_int, _F, _F, c_void_p]),
    "gs_seg_loss_bwd": (c_int, [_F, _P, _F, _F, c_float, _F, c_int, c_int, c_int, c_int, c_void_p]),
    "gs_dice_batched_ws_floats": (c_int64, [c_int]),
    "gs_dice_coeff_batched": (c_int, [_F, _F, c_int, c_int64, _F, _F, c_void_p]),
    "gs_eval_dice": (c_int, [_F, _P, c_int, c_int, c_int64, _F, _F, c_void_p]),
    "gs_jaccard_loss_out_floats": (c_int64, [c_int]),
    "gs_jaccard_seg_loss_fwd": (c_int, [_F, _P, c_int, c_int64, _F, _F, c_void_p]),
    "gs_jaccard_seg_loss_bwd": (c_int, [_F, _P, _F, _F, c_float, _F, c_int, c_int64, c_void_p]),
    "gs_eval_jaccard": (c_int, [_F, _P, c_int, c_int64, _F, _F, c_void_p]),
    "gs_dice_loss_fwd": (c_int, [_F, _F, c_int64, _F, _F, c_void_p]),
    "gs_dice_loss_bwd": (c_int, [_F, _F, _F, _F, c_int64, c_void_p]),
    "gs_mean_loss_fwd": (c_int, [_F, _F, c_float, c_int, c_int64, _F, _F, c_void_p]),
    "gs_pack_weight_split": (c_int, [_F, _P, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gs_pack_weight_segs": (c_int, [c_int, POINTER(GsSegPackDesc), c_int, c_void_p]),
    "gs_pack_weight_q8": (c_int, [c_int, POINTER(GsQ8PackDesc), c_int, c_void_p]),
    "gs_conv3x3_q8_ok": (c_int, [c_int] * 3),
    "gs_conv3x3_q8": (c_int, [_P, _P, _P, _P, _P, _F] + [c_int] * 10 + [c_void_p]),
    "gs_conv3d_3x3x3_q8": (c_int, [_P, _P, _P, _P, _P, _F] + [c_int] * 11 + [c_void_p]),
    "gs_bn_act_apply_split_q8": (c_int, [_P, _P, _F, _F, c_int, _P, _P, c_int, c_int, c_int, _P, _P, c_int, c_int] + [c_int] * 5 + [c_void_p]),
    "gs_stem_fwd_bn_pair_q8": (c_int, [_F, _F, _F, _F, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gs_q8_from_hi": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "gs_conv3d_3x3x3_precise": (c_int, [_P, _P, _P, _P, _F, _F] + [c_int] * 11 + [POINTER(c_int32)] * 3 + [c_int, c_int, c_void_p]),
    "gs_conv3d_3x3x3_precise_to": (c_int, [_P, _P, _P, _P, _F, _F] + [c_int] * 12 + [POINTER(c_int32)] * 3 + [c_int, c_int, c_void_p]),
    "gs_maxpool3d_fwd_pair_q8": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, c_int, c_int] + [c_int] * 6 + [c_void_p]),
    "gs_maxpool3d_fwd_pair": (c_int, [_P, _P, c_int, _P, _P] + [c_int] * 7 + [c_void_p]),
    "gs_upsample2x_bilinear_fwd_pair": (c_int, [_P, _P, _P, _P] + [c_int] * 13 + [c_void_p]),
    "gs_conv3x3_precise": (c_int, [_P, _P, _P, _P, _F, _F] + [c_int] * 10 + [POINTER(c_int32), POINTER(c_int32), c_int, c_int, c_void_p]),
    "gs_upconv2x2_fwd_precise": (c_int, [_P, _P, _F, _P, _P] + [c_int] * 15 + [c_void_p]),
    "gs_conv_smallcin_fwd_split": (c_int, [_F, _F, _P, _P, _F] + [c_int] * 8 + [c_void_p]),
    "gs_bn_act_apply_split": (c_int, [_P, _P, _F, _F, c_int, _P, _P, c_int, c_int, _P, _P] + [c_int] * 6 + [c_void_p]),
    "gs_bn_act_apply_split_pool3d": (c_int, [_P, _P, _F, _F, c_int, _P, _P, c_int, c_int, _P, _P] + [c_int] * 7 + [c_void_p]),
    "gs_head1x1_fwd_split": (c_int, [_P, _P, _F, _F, _F] + [c_int] * 6 + [c_void_p]),
    "gs_head1x1_bn_fwd_split": (c_int, [_P, _P, _F, _F, c_int, _F, _F, _F] + [c_int] * 6 + [c_void_p]),
    "gs_stem_fwd_bn_pair": (c_int, [_F, _F, _F, _F, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gs_stem_bwd_onepass_strided": (c_int, [_F, _P, c_int, _P, c_int, c_int, c_int, _F, _F, c_int, c_int, c_int, c_int, c_void_p]),
    "gs_mean_loss_bwd": (c_int, [_F, _F, c_float, c_int, c_int64, _F, c_float, _F, c_void_p]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into libgsseg_hip.so (in-tree)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libgsseg_hip.so failed:\n" + res.stdout[-4000:])
    return LIB_PATH


def load():
    """Load the shared object and attach the prototypes.  Raises if it is missing -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built.  Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or make -C semantic_segmentation_amd/csrc). "
            "There is no CPU/PyTorch fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)       # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.gs_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libgsseg_hip.so ABI {lib.gs_abi_version()} != binding {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


GS_EUNSUPPORTED = -3          # include/gsseg.h GsStatus


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().gs_last_error()
        raise RuntimeError(f"{what or 'gsseg'} failed (status {rc}): {msg.decode() if msg else ''}")


def call(name: str, *args):
    """Call an int-status entry point and raise RuntimeError with gs_last_error() on failure."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        check(rc, name)
    return rc
