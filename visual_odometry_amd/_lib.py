"""ctypes binding of ``lib/libvo_hip.so`` (the C ABI declared in ``include/vo_hip.h``).

There is no CPU fallback: if the shared library has not been built, or a kernel
is called without a GPU tensor, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_uint64, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libvo_hip.so")

VO_OK, VO_ERR_INVALID_ARG, VO_ERR_LAUNCH, VO_ERR_UNSUPPORTED = 0, -1, -2, -3
ABI_VERSION = 301           # the VO_HIP_VERSION of include/vo_hip.h that SIGNATURES below was written for
VO_PIECES_BF16, VO_PIECES_F16 = 0, 1

_P, _I, _L, _D, _F = c_void_p, c_int, c_int64, c_double, c_float

# name -> argtypes; every function returns int except the two below.
SIGNATURES = {
    "vo_rectify_maps": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _P],
    "vo_remap_bilinear_u8": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vo_erode3x3_u8": [_P, _P, _I, _I, _P],
    "vo_instance_norm_relu": [_P, _P, _L, _L, _F, _I, _I, _P],
    "vo_instance_norm_apply": [_P, _P, _P, _P, _L, _L, _F, _F, _I, _I, _I, _P],
    "vo_channel_affine_relu": [_P, _P, _P, _P, _L, _I, _L, _I, _I, _P],
    "vo_add_relu": [_P, _P, _L, _I, _P],
    "vo_resize_bilinear": [_P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_avg_pool3x3s2": [_P, _P, _L, _I, _I, _I, _P],
    "vo_avg_pool3x3s2_into": [_P, _P, _L, _I, _I, _I, _I, _I, _P],
    "vo_winograd_filter": [_P, _P, _I, _I, _P],
    "vo_conv3x3_winograd": [_P, _P, _P, _L, _P, _P, _I, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P],
    "vo_winograd4_filter": [_P, _P, _I, _I, _P],
    "vo_conv3x3_winograd4": [_P, _P, _P, _L, _P, _P, _I, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P],
    "vo_conv3x3_bf16x3_filter": [_P, _P, _I, _I, _I, _I, _P],
    "vo_conv3x3_bf16x3": [_P, _P, _P, _L, _P, _P, _I, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_conv5_bf16x3_filter": [_P, _P, _I, _I, _I, _I, _P],
    "vo_conv5_bf16x3": [_P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_conv3x3s2_bf16x3": [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_conv3x3s2_shortcut_bf16x3": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_conv1x1_bf16x3_filter": [_P, _P, _I, _I, _I, _I, _P],
    "vo_conv1x1_bf16x3": [_P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_conv1x1_pixel_major_bf16x3": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_conv1x1_split_rows_bf16x3": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vo_conv1x1s2_bf16x3": [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_conv7x7_bf16x3_filter": [_P, _P, _I, _I, _I, _I, _P],
    "vo_conv7x7_bf16x3": [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_gru_conv_zr_bf16x3": [_P, _P, _P, _P, _L, _P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_gru_conv_q_bf16x3": [_P, _P, _P, _P, _P, _L, _P, _P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_winograd1d_filter": [_P, _P, _I, _I, _P],
    "vo_conv5_winograd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vo_conv1x1_filter": [_P, _P, _I, _I, _P],
    "vo_conv1x1": [_P, _P, _P, _L, _P, _P, _I, _I, _I, _L, _I, _P],
    "vo_conv1x1_pixel_major": [_P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _P],
    "vo_conv7x7_filter": [_P, _P, _I, _I, _P],
    "vo_conv7x7": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_conv3x3s2_filter": [_P, _P, _I, _I, _P],
    "vo_conv3x3s2": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vo_conv1x1s2": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vo_shift_add_3x3": [_P, _P, _P, _I, _I, _I, _I, _P],
    "vo_gru_gates": [_P, _P, _P, _P, _L, _P, _L, _P, _L, _P, _P, _L, _I, _I, _L, _I, _P],
    "vo_gru_update": [_P, _P, _P, _L, _P, _P, _L, _P, _L, _P, _L, _I, _I, _L, _I, _P],
    "vo_prepare_images": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_prepare_images_gray": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_corr_volume_2d": [_P, _P, _P, _I, _I, _I, _I, _P],
    "vo_corr_volume_1d": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vo_corr_volume_2d_bf16x3": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "vo_corr_volume_1d_bf16x3": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "vo_feature_rows_amax": [_P, _P, _I, _L, _P],
    "vo_split_feature_rows": [_P, _P, _L, _I, _I, _I, _P, _L, _P],
    "vo_corr_volume_2d_presplit": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "vo_corr_volume_1d_presplit": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "vo_corr_pyramid_2d": [_P, _P, _P, _P, _L, _I, _I, _P],
    "vo_corr_pyramid_1d": [_P, _P, _P, _P, _L, _I, _P],
    "vo_corr_lookup_2d": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P],
    "vo_corr_lookup_1d": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "vo_convex_upsample": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vo_disparity_to_depth": [_P, _P, _L, _D, _I, _P],
    "vo_keypoints_to_3d": [_P, _L, _P, _P, _P, _I, _I, _I, _I, _P],
    "vo_keypoints_flow_2d": [_P, _L, _P, _P, _P, _I, _I, _I, _I, _P],
    "vo_keypoints_flow_3d": [_P, _L, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vo_pair_correspondences": [_P, _L, _P, _P, _P, _P, _D, _D, _D, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vo_keypoints_to_2d": [_P, _P, _P, _P, _L, _P],
    "vo_track_keypoints": [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "vo_kabsch_batch": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vo_pair_poses": [_P, _L, _P, _P, _P, _P, _D, _D, _D, _P, _I, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "vo_pair_poses_static": [_P, _L, _P, _P, _P, _P, _D, _D, _D, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "vo_ransac_kabsch_batch": [_P, _P, _P, _D, _I, c_uint64, _P, _P, _P, _I, _I, _L, _P],
}
# entry points that return a size instead of a status
SIZE_QUERIES = {"vo_pair_poses_workspace_bytes": [_I, _I]}
EXPORTS = sorted(list(SIGNATURES) + list(SIZE_QUERIES) + ["vo_version", "vo_last_error"])

_lib = None


class VoHipError(RuntimeError):
    """A HIP launch was rejected or a shape is outside what the kernels support."""


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m visual_odometry_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.vo_version.argtypes = []
    lib.vo_version.restype = c_int
    if lib.vo_version() != ABI_VERSION:
        raise RuntimeError(
            f"{LIB_PATH} reports version {lib.vo_version()}, this binding was written for {ABI_VERSION}: "
            "rebuild it with `python -m visual_odometry_amd.build` (a stale library would take shifted arguments)")
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_int
    for name, args in SIZE_QUERIES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_int64
    lib.vo_last_error.argtypes = []
    lib.vo_last_error.restype = c_char_p
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    """Invoke an entry point and turn its status into the reference's exceptions:
    invalid arguments -> ValueError, anything else -> VoHipError."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc == VO_OK:
        return
    msg = lib.vo_last_error().decode("utf-8", "replace")
    if rc == VO_ERR_INVALID_ARG:
        raise ValueError(msg)
    raise VoHipError(f"{name} failed ({rc}): {msg}")


def version() -> int:
    return load().vo_version()
