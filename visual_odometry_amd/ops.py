"""Tensor-level wrappers over the C ABI: shape/dtype/device checks on the host,
output allocation through PyTorch's caching allocator, launch on torch's
current stream.  PyTorch is only the memory/stream plumbing here; all the
arithmetic is in ``csrc/*.hip``.

Every function takes and returns CUDA (ROCm) tensors and raises when handed a
CPU tensor: there is deliberately no fallback path.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

MAX_LEVELS = 4


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# Optional per-launch timing: while a recorder dict is installed, every C-ABI call
# is bracketed by two events recorded on the stream the kernel is launched on, so
# a caller can read each kernel's device-side duration after synchronising.
_recorder = None


def set_recorder(rec):
    """Install (dict) or remove (None) the launch recorder; returns the old one.
    The dict maps entry-point name -> list of (start_event, end_event, *sizes)."""
    global _recorder
    old, _recorder = _recorder, rec
    return old


def _call(name: str, *args, nbytes: int = 0, flops: int = 0) -> None:
    """``nbytes`` / ``flops`` = algorithmic bytes / matrix-core FLOPs of this launch (recorded
    for roofline reports)."""
    if _recorder is None:
        _lib.call(name, *args)
        return
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record()
    _lib.call(name, *args)
    end.record()
    _recorder.setdefault(name, []).append((start, end, nbytes, flops))


def _dev(t: torch.Tensor, name: str, dtype: torch.dtype) -> int:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a GPU tensor (the HIP backend has no CPU path)")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def _opt(t, name, dtype) -> int:
    return 0 if t is None else _dev(t, name, dtype)


def _host9(Kinv) -> "ctypes.Array":
    a = np.ascontiguousarray(np.asarray(Kinv, dtype=np.float64)).reshape(-1)
    if a.size != 9:
        raise ValueError("K_inv must have 9 elements")
    return (ctypes.c_double * 9)(*a.tolist())


# ------------------------------------------------------- fused encoder passes

def _host_doubles(a, n=None, name="array"):
    v = np.asarray(a, dtype=np.float64).reshape(-1)
    if n is not None and v.size != n:
        raise ValueError(f"{name} must have {n} elements")
    return (ctypes.c_double * v.size)(*v.tolist())


def rectify_maps(K, D, R, P, height: int, width: int, device) -> tuple:
    """Sampling maps of one camera (cv2.initUndistortRectifyMap, CV_32FC1):
    (map_x, map_y) float32 GPU tensors of shape (height, width)."""
    d = np.asarray(D, dtype=np.float64).reshape(-1)
    if d.size > 8:
        raise ValueError("at most 8 distortion coefficients (k1 k2 p1 p2 k3 k4 k5 k6) are supported")
    mx = torch.empty((height, width), device=device, dtype=torch.float32)
    my = torch.empty_like(mx)
    _call("vo_rectify_maps", _host_doubles(K, 9, "K"), _host_doubles(d) if d.size else None, int(d.size),
          _host_doubles(R, 9, "R"), _host_doubles(np.asarray(P, dtype=np.float64)[:3, :4], 12, "P"),
          mx.data_ptr(), my.data_ptr(), int(height), int(width), _stream())
    return mx, my


def remap_bilinear_u8(frames: torch.Tensor, map_x: torch.Tensor, map_y: torch.Tensor) -> torch.Tensor:
    """(F,H,W) uint8 frames resampled through one (Ho,Wo) map pair -> (F,Ho,Wo) uint8."""
    if frames.dim() != 3:
        raise ValueError("frames must be (F,H,W)")
    if map_x.dim() != 2 or map_x.shape != map_y.shape:
        raise ValueError("map_x and map_y must be (Ho,Wo)")
    F_, H, W = frames.shape
    Ho, Wo = map_x.shape
    out = torch.empty((F_, Ho, Wo), device=frames.device, dtype=torch.uint8)
    _call("vo_remap_bilinear_u8", _dev(frames, "frames", torch.uint8), _dev(map_x, "map_x", torch.float32),
          _dev(map_y, "map_y", torch.float32), out.data_ptr(), F_, H, W, Ho, Wo, _stream(),
          nbytes=F_ * (H * W + Ho * Wo) + 8 * Ho * Wo)
    return out


def erode3x3(mask: torch.Tensor) -> torch.Tensor:
    """(H,W) uint8 0/1 mask -> its 3x3 erosion (outside pixels do not erode)."""
    if mask.dim() != 2:
        raise ValueError("mask must be (H,W)")
    out = torch.empty_like(mask)
    _call("vo_erode3x3_u8", _dev(mask, "mask", torch.uint8), out.data_ptr(), mask.shape[0], mask.shape[1], _stream())
    return out


VO_F32, VO_F16 = 0, 1        # storage codes of the fused conv-net passes (include/vo_hip.h)


def _act(t: torch.Tensor, name: str, like: torch.Tensor | None = None):
    """Contiguous float32 / float16 GPU activation tensor -> (ptr, dtype code, bytes per element)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a GPU tensor (the HIP backend has no CPU path)")
    if t.dtype not in (torch.float32, torch.float16):
        raise ValueError(f"{name}: expected float32 or float16, got {t.dtype}")
    if like is not None and t.dtype != like.dtype:
        raise ValueError(f"{name}: dtype {t.dtype} differs from {like.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t.data_ptr(), (VO_F16 if t.dtype == torch.float16 else VO_F32), t.element_size()


def _skip_ptr(skip, x):
    if skip is None:
        return 0
    if skip.shape != x.shape:
        raise ValueError("skip must have the shape of x")
    return _act(skip, "skip", x)[0]


def instance_norm_relu_(x: torch.Tensor, eps: float = 1e-5, relu: bool = True, skip: torch.Tensor | None = None) -> torch.Tensor:
    """In-place InstanceNorm2d (no affine) [+ ReLU] on an NCHW float32/float16 tensor; with
    ``skip`` the residual join follows in the same pass: x = relu(skip + act(norm(x)))."""
    N, C, H, W = x.shape
    px, dt, es = _act(x, "x")
    _call("vo_instance_norm_relu", px, _skip_ptr(skip, x), N * C, H * W, float(eps), int(relu), dt,
          _stream(), nbytes=(3 + (skip is not None)) * es * x.numel())     # statistics read + apply read(s)/write
    return x


def instance_norm_apply_(x: torch.Tensor, stats: torch.Tensor, eps: float = 1e-5, relu: bool = True,
                         skip: torch.Tensor | None = None, skip_stats: torch.Tensor | None = None,
                         skip_eps: float = 1e-5, skip_relu: bool = False) -> torch.Tensor:
    """instance_norm_relu_ with the plane statistics (N,C,2) float64 (sum, sum of squares)
    already accumulated by the producing convolution: a single read-modify-write pass.  With
    ``skip_stats`` the residual branch is a raw convolution output too and is normalised (and rectified
    with ``skip_relu``) while it is read: x = relu(act_s(norm(skip)) + act(norm(x)))."""
    N, C, H, W = x.shape
    if stats.numel() != N * C * 2:
        raise ValueError("stats must be (N,C,2)")
    if skip_stats is not None and (skip is None or skip_stats.numel() != N * C * 2):
        raise ValueError("skip_stats must be (N,C,2) and come with a skip tensor")
    px, dt, es = _act(x, "x")
    _call("vo_instance_norm_apply", px, _skip_ptr(skip, x), _dev(stats, "stats", torch.float64),
          _opt(skip_stats, "skip_stats", torch.float64), N * C, H * W, float(eps), float(skip_eps), int(relu),
          int(skip_relu), dt, _stream(), nbytes=(2 + (skip is not None)) * es * x.numel())
    return x


def channel_affine_relu_(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, relu: bool = True,
                         skip: torch.Tensor | None = None) -> torch.Tensor:
    """In-place x*scale[c] + shift[c] [+ ReLU] on an NCHW float32/float16 tensor (scale and
    shift float32); with ``skip`` followed by x = relu(skip + x) in the same pass."""
    N, C, H, W = x.shape
    if scale.numel() != C or shift.numel() != C:
        raise ValueError("scale/shift must have one entry per channel")
    px, dt, es = _act(x, "x")
    _call("vo_channel_affine_relu", px, _dev(scale, "scale", torch.float32),
          _dev(shift, "shift", torch.float32), _skip_ptr(skip, x), N, C, H * W, int(relu), dt, _stream(),
          nbytes=(2 + (skip is not None)) * es * x.numel())
    return x


def add_relu_(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """In-place a = relu(a + b)."""
    if a.shape != b.shape:
        raise ValueError("add_relu_: shapes differ")
    pa, dt, es = _act(a, "a")
    _call("vo_add_relu", pa, _act(b, "b", a)[0], a.numel(), dt, _stream(), nbytes=3 * es * a.numel())
    return a


def resize_bilinear(x: torch.Tensor, size, out: torch.Tensor | None = None) -> torch.Tensor:
    """(B,C,h,w) -> (B,C,H,W), bilinear with align_corners=True.  ``out`` may be a channel
    slice of a wider NCHW tensor (dense (C,H,W) blocks, any batch stride)."""
    B, C, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    px, dt, es = _act(x, "x")
    if out is None:
        out = torch.empty((B, C, H, W), device=x.device, dtype=x.dtype)
    elif tuple(out.shape) != (B, C, H, W):
        raise ValueError("out must be (B,C,H,W)")
    po, so = _slice(out, "out", x)
    _call("vo_resize_bilinear", px, po, so, B, C, h, w, H, W, dt, _stream(),
          nbytes=es * (x.numel() + out.numel()))
    return out


def avg_pool3x3s2(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """(B,C,h,w) -> (B,C,(h-1)//2+1,(w-1)//2+1): avg_pool2d(x, 3, stride=2, padding=1).  ``out`` may be a channel
    slice of a wider NCHW tensor (dense (C,H,W) blocks, any batch stride)."""
    B, C, h, w = x.shape
    H, W = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    px, dt, es = _act(x, "x")
    if out is None:
        out = torch.empty((B, C, H, W), device=x.device, dtype=x.dtype)
        _call("vo_avg_pool3x3s2", px, out.data_ptr(), B * C, h, w, dt, _stream(), nbytes=es * (x.numel() + out.numel()))
        return out
    if tuple(out.shape) != (B, C, H, W):
        raise ValueError("out must be (B,C,H,W)")
    po, so = _slice(out, "out", x)
    _call("vo_avg_pool3x3s2_into", px, po, so, B, C, h, w, dt, _stream(), nbytes=es * (x.numel() + out.numel()))
    return out


# ------------------------------------------------------------ split arithmetic
# The split convolutions cut every float32 operand into two 16-bit pieces; ``pieces`` names the format:
# "bf16" (16 significant bits, float32's exponent range) or "f16" (22 significant bits; float16's exponent
# range is met with exact powers of two -- see include/vo_hip.h, VO_PIECES_F16).  An arranged weight tensor
# carries its format and its power of two as attributes, so a convolution cannot be handed the wrong ones.
PIECES = {"bf16": _lib.VO_PIECES_BF16, "f16": _lib.VO_PIECES_F16}


def _weight_exp(weight: torch.Tensor, pieces: str) -> int:
    """The e with max|w| * 2^e in [2^13, 2^14) for float16 pieces (0 for bf16, and for an all-zero weight)."""
    if pieces != "f16":
        return 0
    m = float(weight.detach().abs().max())
    if not np.isfinite(m):
        raise ValueError("weights must be finite")
    if m == 0.0:
        return 0
    return int(13 - np.floor(np.log2(m)))


def _tag(arranged: torch.Tensor, pieces: str, wexp: int) -> torch.Tensor:
    arranged.vo_pieces, arranged.vo_wexp = pieces, wexp
    return arranged


def _pieces_of(arranged: torch.Tensor):
    """(pieces code, weight_exp) of an arranged weight tensor (bf16 pieces when it carries no tag)."""
    return PIECES[getattr(arranged, "vo_pieces", "bf16")], int(getattr(arranged, "vo_wexp", 0))


def winograd_filter(weight: torch.Tensor, tile: int = 2) -> torch.Tensor:
    """(K,C,3,3) float32 weight -> the transformed filter conv3x3_winograd consumes.
    ``tile`` 2: F(2x2,3x3), 16 positions; 4: F(4x4,3x3), 36 positions (fewer multiplies,
    rounding error about 1e-5 of the largest output instead of 5e-7)."""
    if weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3):
        raise ValueError("weight must be (K,C,3,3)")
    if tile not in (2, 4):
        raise ValueError("tile must be 2 or 4")
    K, C = weight.shape[:2]
    if C % 8:
        raise ValueError("the Winograd path needs a multiple of 8 input channels")
    positions = 16 if tile == 2 else 36
    out = torch.empty(((K + 63) // 64 * 64 * C * positions,), device=weight.device, dtype=torch.float32)
    _call("vo_winograd_filter" if tile == 2 else "vo_winograd4_filter", _dev(weight.detach(), "weight", torch.float32),
          out.data_ptr(), K, C, _stream())
    return out


def conv3x3_split_filter(weight: torch.Tensor, pieces: str = "bf16") -> torch.Tensor:
    """(K,C,3,3) float32 weight -> its high/low 16-bit parts in operand order that conv3x3_winograd
    hands to the split direct kernel (a float32 tensor used as raw storage, ceil(K/64)*64*C*9
    elements, tagged with ``pieces`` and its power of two).  C must be a multiple of 16."""
    if weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3):
        raise ValueError("weight must be (K,C,3,3)")
    K, C = weight.shape[:2]
    if C % 16:
        raise ValueError("the split-bf16 path needs a multiple of 16 input channels")
    out = torch.empty(((K + 63) // 64 * 64 * C * 9,), device=weight.device, dtype=torch.float32)
    wexp = _weight_exp(weight, pieces)
    _call("vo_conv3x3_bf16x3_filter", _dev(weight.detach().contiguous(), "weight", torch.float32), out.data_ptr(), K, C,
          PIECES[pieces], wexp, _stream())
    return _tag(out, pieces, wexp)


def conv3x3_winograd(x: torch.Tensor, transformed: torch.Tensor, K: int, scale: torch.Tensor | None = None,
                     shift: torch.Tensor | None = None, relu: bool = False,
                     skip: torch.Tensor | None = None, out: torch.Tensor | None = None,
                     stats: torch.Tensor | None = None, in_stats: torch.Tensor | None = None,
                     in_eps: float = 1e-5) -> torch.Tensor:
    """x (N,C,H,W) float32, transformed filter of a (K,C,3,3) weight -> conv2d(x, weight, padding=1),
    optionally followed in the same kernel by *scale[k] + shift[k], ReLU and relu(. + skip).
    The size of ``transformed`` tells which of the two kernels it was made for (winograd_filter's tile).
    ``out`` may be a channel slice of a wider NCHW tensor (dense (K,H,W) blocks, any batch stride).
    ``stats``: zeroed float64 (N,K,2) that receives each plane's sum and sum of squares of the
    raw output (for instance_norm_apply_; no epilogue in that case).
    ``in_stats``: such sums of the INPUT x (N,C,2): x is instance-normalised and ReLU-ed while it
    is loaded, so the tensor between two convolutions needs no normalisation pass of its own."""
    N, C, H, W = x.shape
    per_position = (K + 63) // 64 * 64 * C
    if transformed.numel() == per_position * 16:
        name, tile, positions = "vo_conv3x3_winograd", 2, 16
    elif transformed.numel() == per_position * 36:
        name, tile, positions = "vo_conv3x3_winograd4", 4, 36
    elif transformed.numel() == per_position * 9:
        # split-bf16 direct kernel: three bf16 matrix instructions per product, 9 taps
        name, tile, positions = "vo_conv3x3_bf16x3", 1, 27
        if W % 2:
            raise ValueError("the split-bf16 kernel needs an even width")
    else:
        raise ValueError("transformed filter does not match (K, C)")
    for t, tname in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{tname} must have one entry per output channel")
    if scale is not None and shift is None:
        raise ValueError("scale needs shift")
    if skip is not None and tuple(skip.shape) != (N, K, H, W):
        raise ValueError("skip must have the shape of the output")
    if out is None:
        out = torch.empty((N, K, H, W), device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (N, K, H, W) or out.dtype != torch.float32:
        raise ValueError("out must be float32 (N,K,H,W)")
    py, sy = _slice(out, "out")
    if in_stats is not None and in_stats.numel() != N * C * 2:
        raise ValueError("in_stats must be (N,C,2) float64")
    if stats is not None:
        if stats.numel() != N * K * 2 or shift is not None or relu or skip is not None:
            raise ValueError("stats must be (N,K,2) float64 and excludes the epilogue")
    _call(name, _dev(x, "x", torch.float32), _dev(transformed, "filter", torch.float32), py, sy,
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu),
          _opt(skip, "skip", torch.float32), _opt(stats, "stats", torch.float64),
          _opt(in_stats, "in_stats", torch.float64), float(in_eps), N, C, H, W, K,
          *(_pieces_of(transformed) if tile == 1 else ()), _stream(),
          nbytes=4 * (x.numel() + out.numel() * (1 + (skip is not None))),
          flops=2 * positions * C * K * N * ((H + tile - 1) // tile) * ((W + tile - 1) // tile))   # multiply-adds the kernel issues
                                                                                                  # (Winograd domain / three bf16 terms)
    return out


def winograd1d_filter(weight: torch.Tensor) -> torch.Tensor:
    """(K,C,1,5) or (K,C,5,1) float32 weight -> the transformed filter conv5_winograd consumes."""
    if weight.dim() != 4 or tuple(weight.shape[2:]) not in ((1, 5), (5, 1)):
        raise ValueError("weight must be (K,C,1,5) or (K,C,5,1)")
    K, C = weight.shape[:2]
    if C % 8:
        raise ValueError("the Winograd path needs a multiple of 8 input channels")
    out = torch.empty(((K + 63) // 64 * 64 * C * 8,), device=weight.device, dtype=torch.float32)
    _call("vo_winograd1d_filter", _dev(weight.detach().contiguous(), "weight", torch.float32), out.data_ptr(), K, C, _stream())
    return out


def conv5_winograd(x: torch.Tensor, transformed: torch.Tensor, K: int, vertical: bool) -> torch.Tensor:
    """x (N,C,H,W) float32, transformed filter of a (K,C,1,5) [vertical False] or (K,C,5,1) [True]
    weight -> the stride-1 "same" convolution conv2d(x, weight, padding=(0,2) / (2,0))."""
    N, C, H, W = x.shape
    if transformed.numel() != (K + 63) // 64 * 64 * C * 8:
        raise ValueError("transformed filter does not match (K, C)")
    out = torch.empty((N, K, H, W), device=x.device, dtype=torch.float32)
    _call("vo_conv5_winograd", _dev(x, "x", torch.float32), _dev(transformed, "filter", torch.float32), out.data_ptr(),
          N, C, H, W, K, int(bool(vertical)), _stream(), nbytes=4 * (x.numel() + out.numel()),
          flops=2 * 8 * C * K * N * (H * ((W + 3) // 4) if not vertical else ((H + 3) // 4) * W))   # transformed-domain multiply-adds
    return out


def conv5_split_filter(weight: torch.Tensor, pieces: str = "bf16") -> torch.Tensor:
    """(K,C,1,5) or (K,C,5,1) float32 weight -> bf16 high/low parts in operand order for conv5_split
    (a float32 tensor used as raw storage).  C must be a multiple of 16."""
    if weight.dim() != 4 or tuple(weight.shape[2:]) not in ((1, 5), (5, 1)):
        raise ValueError("weight must be (K,C,1,5) or (K,C,5,1)")
    K, C = weight.shape[:2]
    if C % 16:
        raise ValueError("the split-bf16 path needs a multiple of 16 input channels")
    out = torch.empty(((K + 63) // 64 * 64 * C * 5,), device=weight.device, dtype=torch.float32)
    wexp = _weight_exp(weight, pieces)
    _call("vo_conv5_bf16x3_filter", _dev(weight.detach().contiguous(), "weight", torch.float32), out.data_ptr(), K, C,
          PIECES[pieces], wexp, _stream())
    return _tag(out, pieces, wexp)


def conv5_split(x: torch.Tensor, arranged: torch.Tensor, K: int, vertical: bool, scale: torch.Tensor | None = None,
                shift: torch.Tensor | None = None, relu: bool = False, out: torch.Tensor | None = None) -> torch.Tensor:
    """conv5_winograd's convolution (1x5 / 5x1, stride 1, "same") on the split-bf16 direct kernel, with
    an optional fused v*scale[k] + shift[k] / ReLU and a channel-slice output."""
    N, C, H, W = x.shape
    if arranged.numel() != (K + 63) // 64 * 64 * C * 5:
        raise ValueError("arranged weights do not match (K, C)")
    if W % 2:
        raise ValueError("the split-bf16 kernel needs an even width")
    for t, name in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{name} must have one entry per output channel")
    if out is None:
        out = torch.empty((N, K, H, W), device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (N, K, H, W) or out.dtype != torch.float32:
        raise ValueError("out must be float32 (N,K,H,W)")
    py, sy = _slice(out, "out")
    _call("vo_conv5_bf16x3", _dev(x, "x", torch.float32), _dev(arranged, "weights", torch.float32), py, sy,
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu), N, C, H, W, K,
          int(bool(vertical)), *_pieces_of(arranged), _stream(), nbytes=4 * (x.numel() + out.numel()), flops=2 * 15 * N * K * C * H * W)
    return out


def conv3x3s2_split(x: torch.Tensor, arranged: torch.Tensor, K: int, scale: torch.Tensor | None = None,
                    shift: torch.Tensor | None = None, relu: bool = False, stats: torch.Tensor | None = None) -> torch.Tensor:
    """conv2d(x, weight, stride=2, padding=1) on the split-bf16 direct kernel (weights from
    conv3x3_split_filter), optional epilogue, or ``stats`` (zeroed float64 (N,K,2)) receiving the plane
    sums of the raw output."""
    N, C, H, W = x.shape
    if arranged.numel() != (K + 63) // 64 * 64 * C * 9:
        raise ValueError("arranged weights do not match (K, C)")
    if W % 2:
        raise ValueError("the split-bf16 kernel needs an even width")
    for t, name in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{name} must have one entry per output channel")
    if stats is not None and (stats.numel() != N * K * 2 or shift is not None or relu):
        raise ValueError("stats must be (N,K,2) float64 and excludes the epilogue")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, K, Ho, Wo), device=x.device, dtype=torch.float32)
    _call("vo_conv3x3s2_bf16x3", _dev(x, "x", torch.float32), _dev(arranged, "weights", torch.float32), out.data_ptr(),
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu),
          _opt(stats, "stats", torch.float64), N, C, H, W, K, *_pieces_of(arranged), _stream(),
          nbytes=4 * (x.numel() + out.numel()), flops=2 * 27 * N * K * C * Ho * Wo)
    return out


def conv1x1_split_filter(weight: torch.Tensor, pieces: str = "bf16") -> torch.Tensor:
    """(K,C,1,1) or (K,C) float32 weight, C a multiple of 16 -> the bf16 high/low parts in operand order
    that conv1x1s2_split consumes (a float32 tensor used as raw storage, ceil(K/64)*64*C elements)."""
    if weight.dim() not in (2, 4) or (weight.dim() == 4 and tuple(weight.shape[2:]) != (1, 1)):
        raise ValueError("weight must be (K,C,1,1) or (K,C)")
    K, C = weight.shape[:2]
    if C % 16:
        raise ValueError("the split-bf16 path needs a multiple of 16 input channels")
    out = torch.empty(((K + 63) // 64 * 64 * C,), device=weight.device, dtype=torch.float32)
    wexp = _weight_exp(weight, pieces)
    _call("vo_conv1x1_bf16x3_filter", _dev(weight.detach().contiguous(), "weight", torch.float32), out.data_ptr(), K, C,
          PIECES[pieces], wexp, _stream())
    return _tag(out, pieces, wexp)


def conv1x1_split(x: torch.Tensor, arranged: torch.Tensor, K: int, scale: torch.Tensor | None = None,
                  shift: torch.Tensor | None = None, relu: bool = False, out: torch.Tensor | None = None,
                  use_template: bool = False) -> torch.Tensor:
    """conv2d(x, weight(K,C,1,1)) on the split-bf16 direct kernel (weights from conv1x1_split_filter), optional
    epilogue v*scale[k] + shift[k], ReLU; ``out``: a (N,K,H,W) channel slice that receives the result."""
    N, C, H, W = x.shape
    if arranged.numel() != (K + 63) // 64 * 64 * C:
        raise ValueError("arranged weights do not match (K, C)")
    if W % 2:
        raise ValueError("the split-bf16 kernel needs an even width")
    for t, name in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{name} must have one entry per output channel")
    if out is None:
        out = torch.empty((N, K, H, W), device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (N, K, H, W) or out.dtype != torch.float32:
        raise ValueError("out must be a float32 (N,K,H,W) tensor or channel slice")
    py, sy = _slice(out, "out")
    _call("vo_conv1x1_bf16x3", _dev(x, "x", torch.float32), _dev(arranged, "weights", torch.float32), py, sy,
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu), N, C, H, W, K,
          *_pieces_of(arranged), int(bool(use_template)), _stream(),
          nbytes=4 * (x.numel() + out.numel()), flops=2 * 3 * N * K * C * H * W)
    return out


def conv1x1_pixel_major_split(x: torch.Tensor, arranged: torch.Tensor, K: int, scale: torch.Tensor | None = None,
                              shift: torch.Tensor | None = None, relu: bool = False, split_rows: bool = False) -> torch.Tensor:
    """conv1x1_split with the output in pixel-major order: x (N,C,H,W) float32 -> (N,H,W,K), K a multiple of 4.
    ``split_rows``: the rows leave as bf16 pieces in blocks of 32 channels (the form of split_feature_rows(., 32),
    K a multiple of 32): bytes for corr_volume_2d_presplit, not float32 numbers."""
    N, C, H, W = x.shape
    if arranged.numel() != (K + 63) // 64 * 64 * C or K % (32 if split_rows else 4):
        raise ValueError("arranged weights do not match (K, C), K a multiple of 4 (of 32 for split rows)")
    if W % 2:
        raise ValueError("the split-bf16 kernel needs an even width")
    for t, name in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{name} must have one entry per output channel")
    out = torch.empty((N, H, W, K), device=x.device, dtype=torch.float32)
    if split_rows and _pieces_of(arranged)[0] != PIECES["bf16"]:
        raise ValueError("rows split beforehand are bf16 pieces: the weights must be arranged with pieces='bf16'")
    _call("vo_conv1x1_split_rows_bf16x3" if split_rows else "vo_conv1x1_pixel_major_bf16x3", _dev(x, "x", torch.float32),
          _dev(arranged, "weights", torch.float32), out.data_ptr(),
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu), N, C, H, W, K,
          *(() if split_rows else _pieces_of(arranged)), _stream(),
          nbytes=4 * (x.numel() + out.numel()), flops=2 * 3 * N * K * C * H * W)
    return out


def _both_pieces(a: torch.Tensor, b: torch.Tensor):
    (pa, ea), (pb, eb) = _pieces_of(a), _pieces_of(b)
    if pa != pb:
        raise ValueError("the two weight tensors were arranged for different piece formats")
    return pa, ea, eb


def conv3x3s2_shortcut_split(x: torch.Tensor, arranged: torch.Tensor, arranged1x1: torch.Tensor, K: int,
                             scale: torch.Tensor | None = None, shift: torch.Tensor | None = None, relu: bool = False,
                             stats: torch.Tensor | None = None, scale2: torch.Tensor | None = None,
                             shift2: torch.Tensor | None = None, relu2: bool = False, stats2: torch.Tensor | None = None):
    """A downsampling residual block's 3x3 stride-2 convolution (padding 1) and its 1x1 stride-2 projection shortcut in
    ONE pass over x (N,C,H,W) float32: the shortcut's input samples are the 3x3 filter's centre tap, so the staged
    patch serves both.  Weights from conv3x3_split_filter and conv1x1_split_filter, K output channels each; per
    output scale / shift / relu, or ``stats`` (N,K,2) float64 that receive the raw output's plane sums.  Needs an
    output width in multiples of 4.  Returns (y, shortcut), both (N,K,ceil(H/2),ceil(W/2))."""
    N, C, H, W = x.shape
    if arranged.numel() != (K + 63) // 64 * 64 * C * 9 or arranged1x1.numel() != (K + 63) // 64 * 64 * C:
        raise ValueError("arranged weights do not match (K, C)")
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    if W % 2 or Wo % 4:
        raise ValueError("the fused stride-2 kernel needs an even width and an output width in multiples of 4")
    for t, name in ((scale, "scale"), (shift, "shift"), (scale2, "scale2"), (shift2, "shift2")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{name} must have one entry per output channel")
    for t, name in ((stats, "stats"), (stats2, "stats2")):
        if t is not None and (tuple(t.shape) != (N, K, 2) or t.dtype != torch.float64):
            raise ValueError(f"{name} must be a float64 (N,K,2) tensor")
    y = torch.empty((N, K, Ho, Wo), device=x.device, dtype=torch.float32)
    y2 = torch.empty_like(y)
    _call("vo_conv3x3s2_shortcut_bf16x3", _dev(x, "x", torch.float32), _dev(arranged, "weights", torch.float32),
          _dev(arranged1x1, "shortcut weights", torch.float32), y.data_ptr(), y2.data_ptr(),
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu), _opt(stats, "stats", torch.float64),
          _opt(scale2, "scale2", torch.float32), _opt(shift2, "shift2", torch.float32), int(relu2), _opt(stats2, "stats2", torch.float64),
          N, C, H, W, K, *_both_pieces(arranged, arranged1x1), _stream(),
          nbytes=4 * (x.numel() + 2 * y.numel()), flops=2 * 3 * N * K * C * Ho * Wo * 10)
    return y, y2


def conv1x1s2_split(x: torch.Tensor, arranged: torch.Tensor, K: int, scale: torch.Tensor | None = None,
                    shift: torch.Tensor | None = None, relu: bool = False, stats: torch.Tensor | None = None) -> torch.Tensor:
    """conv2d(x, weight(K,C,1,1), stride=2) on the split-bf16 direct kernel (weights from
    conv1x1_split_filter), optional epilogue, or ``stats`` (zeroed float64 (N,K,2)) receiving the plane
    sums of the raw output."""
    N, C, H, W = x.shape
    if arranged.numel() != (K + 63) // 64 * 64 * C:
        raise ValueError("arranged weights do not match (K, C)")
    if W % 2:
        raise ValueError("the split-bf16 kernel needs an even width")
    for t, name in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{name} must have one entry per output channel")
    if stats is not None and (stats.numel() != N * K * 2 or shift is not None or relu):
        raise ValueError("stats must be (N,K,2) float64 and excludes the epilogue")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, K, Ho, Wo), device=x.device, dtype=torch.float32)
    _call("vo_conv1x1s2_bf16x3", _dev(x, "x", torch.float32), _dev(arranged, "weights", torch.float32), out.data_ptr(),
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu),
          _opt(stats, "stats", torch.float64), N, C, H, W, K, *_pieces_of(arranged), _stream(),
          nbytes=4 * (x.numel() // 2 + out.numel()), flops=2 * 3 * N * K * C * Ho * Wo)
    return out


def conv7x7_split_filter(weight: torch.Tensor, pieces: str = "bf16") -> torch.Tensor:
    """(K,C,7,7) float32 weight, C = 1 or 2 -> the bf16 high/low parts in operand order that conv7x7_split
    consumes (a float32 tensor used as raw storage, ceil(K/64)*64*C*64 elements)."""
    if weight.dim() != 4 or tuple(weight.shape[2:]) != (7, 7) or weight.shape[1] not in (1, 2):
        raise ValueError("weight must be (K,C,7,7) with one or two input channels")
    K, C = weight.shape[:2]
    out = torch.empty(((K + 63) // 64 * 64 * C * 64,), device=weight.device, dtype=torch.float32)
    wexp = _weight_exp(weight, pieces)
    _call("vo_conv7x7_bf16x3_filter", _dev(weight.detach().contiguous(), "weight", torch.float32), out.data_ptr(), K, C,
          PIECES[pieces], wexp, _stream())
    return _tag(out, pieces, wexp)


def conv7x7_split(x: torch.Tensor, arranged: torch.Tensor, K: int, scale: torch.Tensor | None = None,
                  shift: torch.Tensor | None = None, relu: bool = False, stride: int = 1,
                  stats: torch.Tensor | None = None) -> torch.Tensor:
    """x (N,C,H,W) float32, C = 1 or 2 -> conv2d(x, weight, stride=stride, padding=3) with split-bf16 products,
    then [* scale[k] + shift[k]] [ReLU], or ``stats`` (zeroed float64 (N,K,2)) receiving the plane sums of the
    raw output."""
    N, C, H, W = x.shape
    if stride not in (1, 2):
        raise ValueError("stride must be 1 or 2")
    if C not in (1, 2) or arranged.numel() != (K + 63) // 64 * 64 * C * 64:
        raise ValueError("arranged weights do not match (K, C), C = 1 or 2")
    for t, name in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{name} must have one entry per output channel")
    if stats is not None and (stats.numel() != N * K * 2 or shift is not None or relu):
        raise ValueError("stats must be (N,K,2) float64 and excludes the epilogue")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((N, K, Ho, Wo), device=x.device, dtype=torch.float32)
    _call("vo_conv7x7_bf16x3", _dev(x, "x", torch.float32), _dev(arranged, "weights", torch.float32), out.data_ptr(),
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu),
          _opt(stats, "stats", torch.float64), N, C, H, W, K, stride, *_pieces_of(arranged), _stream(),
          nbytes=4 * (x.numel() + out.numel()), flops=2 * 3 * N * K * C * 64 * Ho * Wo)
    return out


def conv1x1_filter(weight: torch.Tensor) -> torch.Tensor:
    """(K,C,1,1) or (K,C) float32 weight -> the arranged weights conv1x1 consumes."""
    if weight.dim() not in (2, 4) or (weight.dim() == 4 and tuple(weight.shape[2:]) != (1, 1)):
        raise ValueError("weight must be (K,C,1,1) or (K,C)")
    K, C = weight.shape[:2]
    out = torch.empty(((K + 63) // 64 * 64 * ((C + 7) // 8 * 8),), device=weight.device, dtype=torch.float32)
    _call("vo_conv1x1_filter", _dev(weight.detach().contiguous(), "weight", torch.float32), out.data_ptr(), K, C, _stream())
    return out


def conv1x1(x: torch.Tensor, arranged: torch.Tensor, K: int, scale: torch.Tensor | None = None,
            shift: torch.Tensor | None = None, relu: bool = False, out: torch.Tensor | None = None) -> torch.Tensor:
    """x (N,C,H,W) float32, arranged (K,C) weights -> conv2d(x, weight) [* scale[k] + shift[k]] [ReLU].
    ``out`` may be a channel slice of a wider NCHW tensor."""
    N, C, H, W = x.shape
    if arranged.numel() != (K + 63) // 64 * 64 * ((C + 7) // 8 * 8):
        raise ValueError("arranged weights do not match (K, C)")
    for t, name in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{name} must have one entry per output channel")
    if out is None:
        out = torch.empty((N, K, H, W), device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (N, K, H, W) or out.dtype != torch.float32:
        raise ValueError("out must be float32 (N,K,H,W)")
    py, sy = _slice(out, "out")
    _call("vo_conv1x1", _dev(x, "x", torch.float32), _dev(arranged, "weights", torch.float32), py, sy,
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu), N, C, H * W, K, _stream(),
          nbytes=4 * (x.numel() + out.numel()), flops=2 * N * K * C * H * W)
    return out


def conv1x1_pixel_major(x: torch.Tensor, arranged: torch.Tensor, K: int, scale: torch.Tensor | None = None,
                        shift: torch.Tensor | None = None, relu: bool = False) -> torch.Tensor:
    """conv1x1 with the output in pixel-major order: x (N,C,H,W) float32 -> (N,H,W,K), i.e.
    conv2d(x, weight).permute(0, 2, 3, 1).contiguous() without the copy (the layout the correlation
    kernels read)."""
    N, C, H, W = x.shape
    if arranged.numel() != (K + 63) // 64 * 64 * ((C + 7) // 8 * 8):
        raise ValueError("arranged weights do not match (K, C)")
    for t, name in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{name} must have one entry per output channel")
    out = torch.empty((N, H, W, K), device=x.device, dtype=torch.float32)
    _call("vo_conv1x1_pixel_major", _dev(x, "x", torch.float32), _dev(arranged, "weights", torch.float32), out.data_ptr(),
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu), N, C, H * W, K, _stream(),
          nbytes=4 * (x.numel() + out.numel()), flops=2 * N * K * C * H * W)
    return out


def conv7x7_filter(weight: torch.Tensor) -> torch.Tensor:
    """(K,C,7,7) float32 weight, C <= 8 -> the arranged weights conv7x7 consumes."""
    if weight.dim() != 4 or tuple(weight.shape[2:]) != (7, 7) or weight.shape[1] > 8:
        raise ValueError("weight must be (K,C,7,7) with at most 8 input channels")
    K, C = weight.shape[:2]
    out = torch.empty(((K + 63) // 64 * 64 * C * 56,), device=weight.device, dtype=torch.float32)
    _call("vo_conv7x7_filter", _dev(weight.detach().contiguous(), "weight", torch.float32), out.data_ptr(), K, C, _stream())
    return out


def conv7x7(x: torch.Tensor, arranged: torch.Tensor, K: int, scale: torch.Tensor | None = None,
            shift: torch.Tensor | None = None, relu: bool = False, stride: int = 1) -> torch.Tensor:
    """x (N,C,H,W) float32, C <= 8 -> conv2d(x, weight, stride=stride, padding=3) [* scale[k] + shift[k]] [ReLU];
    stride 1 or 2."""
    N, C, H, W = x.shape
    if stride not in (1, 2):
        raise ValueError("stride must be 1 or 2")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if arranged.numel() != (K + 63) // 64 * 64 * C * 56:
        raise ValueError("arranged weights do not match (K, C)")
    for t, name in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{name} must have one entry per output channel")
    out = torch.empty((N, K, Ho, Wo), device=x.device, dtype=torch.float32)
    _call("vo_conv7x7", _dev(x, "x", torch.float32), _dev(arranged, "weights", torch.float32), out.data_ptr(),
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu), N, C, H, W, K, stride,
          _stream(), nbytes=4 * (x.numel() + out.numel()), flops=2 * N * K * C * 49 * Ho * Wo)
    return out


def conv3x3s2_filter(weight: torch.Tensor) -> torch.Tensor:
    """(K,C,3,3) float32 weight, C a multiple of 8 -> the arranged weights conv3x3s2 consumes."""
    if weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or weight.shape[1] % 8:
        raise ValueError("weight must be (K,C,3,3) with a multiple of 8 input channels")
    K, C = weight.shape[:2]
    out = torch.empty(((K + 63) // 64 * 64 * C * 9,), device=weight.device, dtype=torch.float32)
    _call("vo_conv3x3s2_filter", _dev(weight.detach().contiguous(), "weight", torch.float32), out.data_ptr(), K, C, _stream())
    return out


def conv3x3s2(x: torch.Tensor, arranged: torch.Tensor, K: int, scale: torch.Tensor | None = None,
              shift: torch.Tensor | None = None, relu: bool = False) -> torch.Tensor:
    """x (N,C,H,W) float32 -> conv2d(x, weight, stride=2, padding=1) [* scale[k] + shift[k]] [ReLU]."""
    N, C, H, W = x.shape
    if arranged.numel() != (K + 63) // 64 * 64 * C * 9:
        raise ValueError("arranged weights do not match (K, C)")
    for t, name in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{name} must have one entry per output channel")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, K, Ho, Wo), device=x.device, dtype=torch.float32)
    _call("vo_conv3x3s2", _dev(x, "x", torch.float32), _dev(arranged, "weights", torch.float32), out.data_ptr(),
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu), N, C, H, W, K, _stream(),
          nbytes=4 * (x.numel() + out.numel()), flops=2 * N * K * C * 9 * Ho * Wo)
    return out


def conv1x1s2(x: torch.Tensor, arranged: torch.Tensor, K: int, scale: torch.Tensor | None = None,
              shift: torch.Tensor | None = None, relu: bool = False) -> torch.Tensor:
    """x (N,C,H,W) float32, C a multiple of 8, weights from conv1x1_filter -> conv2d(x, weight, stride=2)
    [* scale[k] + shift[k]] [ReLU]: the projection shortcut, reading only the pixels it samples."""
    N, C, H, W = x.shape
    if C % 8 or arranged.numel() != (K + 63) // 64 * 64 * C:
        raise ValueError("arranged weights do not match (K, C), or C is no multiple of 8")
    for t, name in ((scale, "scale"), (shift, "shift")):
        if t is not None and t.numel() != K:
            raise ValueError(f"{name} must have one entry per output channel")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, K, Ho, Wo), device=x.device, dtype=torch.float32)
    _call("vo_conv1x1s2", _dev(x, "x", torch.float32), _dev(arranged, "weights", torch.float32), out.data_ptr(),
          _opt(scale, "scale", torch.float32), _opt(shift, "shift", torch.float32), int(relu), N, C, H, W, K, _stream(),
          nbytes=4 * (x.numel() // 4 + out.numel()), flops=2 * N * K * C * Ho * Wo)
    return out


def shift_add_3x3(planes: torch.Tensor, bias: torch.Tensor | None, K: int) -> torch.Tensor:
    """planes (N, K*9, H, W): per-tap 1x1 products in (k, r, s) order -> the 3x3 / padding 1
    convolution output (N, K, H, W) (+ bias)."""
    N, K9, H, W = planes.shape
    if K9 != 9 * K or (bias is not None and bias.numel() != K):
        raise ValueError("planes must be (N, 9K, H, W) and bias (K)")
    out = torch.empty((N, K, H, W), device=planes.device, dtype=torch.float32)
    _call("vo_shift_add_3x3", _dev(planes, "planes", torch.float32), _opt(bias, "bias", torch.float32), out.data_ptr(),
          N, K, H, W, _stream(), nbytes=4 * (planes.numel() + out.numel()))
    return out


def _slice(t, name: str, like: torch.Tensor | None = None):
    """A (B,C,H,W) GPU tensor whose (C,H,W) block is dense but whose batch items may be
    further apart (a channel slice of a bigger NCHW tensor) -> (ptr, batch stride)."""
    if t is None:
        return 0, 0
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dim() != 4 or t.dtype not in (torch.float32, torch.float16):
        raise ValueError(f"{name}: expected a 4-D float32/float16 GPU tensor")
    if like is not None and t.dtype != like.dtype:
        raise ValueError(f"{name}: dtype {t.dtype} differs from {like.dtype}")
    B, C, H, W = t.shape
    st = t.stride()
    if not (st[3] == 1 and st[2] == W and st[1] == H * W):
        raise ValueError(f"{name}: channel planes must be dense (got strides {st})")
    return t.data_ptr(), (st[0] if B > 1 else C * H * W)


def gru_gates(zr: torch.Tensor, bias_z, bias_r, cz, cr, h: torch.Tensor, rh_out: torch.Tensor) -> torch.Tensor:
    """z = sigmoid(zr[:, :C] + bias_z + cz) (returned), rh_out <- sigmoid(zr[:, C:] + bias_r + cr) * h."""
    B, C2, H, W = zr.shape
    C = C2 // 2
    if tuple(h.shape) != (B, C, H, W) or tuple(rh_out.shape) != (B, C, H, W):
        raise ValueError("gru_gates: h / rh_out must be (B,C,H,W) with zr (B,2C,H,W)")
    pzr, dt, es = _act(zr, "zr")
    z = torch.empty((B, C, H, W), device=zr.device, dtype=zr.dtype)
    pcz, scz = _slice(cz, "cz", zr); pcr, scr = _slice(cr, "cr", zr); ph, sh = _slice(h, "h", zr)
    prh, srh = _slice(rh_out, "rh_out", zr)
    _call("vo_gru_gates", pzr, _opt(bias_z, "bias_z", torch.float32),
          _opt(bias_r, "bias_r", torch.float32), pcz, scz, pcr, scr, ph, sh, z.data_ptr(), prh, srh, B, C, H * W, dt,
          _stream(), nbytes=es * z.numel() * (5 + (cz is not None) + (cr is not None)))
    return z


def gru_update(q: torch.Tensor, bias_q, cq, z: torch.Tensor, h: torch.Tensor, h_out: torch.Tensor, h_copy=None):
    """h_out <- (1-z)*h + z*tanh(q + bias_q + cq); h_copy (optional slice) gets the same values."""
    B, C, H, W = q.shape
    for t, n in ((z, "z"), (h, "h"), (h_out, "h_out")):
        if tuple(t.shape) != (B, C, H, W):
            raise ValueError(f"gru_update: {n} must be (B,C,H,W)")
    pq, dt, es = _act(q, "q")
    pcq, scq = _slice(cq, "cq", q); ph, sh = _slice(h, "h", q); po, so = _slice(h_out, "h_out", q)
    pc, sc = _slice(h_copy, "h_copy", q)
    _call("vo_gru_update", pq, _opt(bias_q, "bias_q", torch.float32), pcq, scq,
          _act(z, "z", q)[0], ph, sh, po, so, pc, sc, B, C, H * W, dt, _stream(),
          nbytes=es * q.numel() * (4 + (cq is not None) + (h_copy is not None)))
    return h_out


_GRU_SHAPES = {(3, 3): 0, (1, 5): 1, (5, 1): 2}


def gru_split_filter(weight: torch.Tensor, pieces: str = "bf16") -> torch.Tensor:
    """(K,Cin,kh,kw) float32 GRU gate weights with (kh,kw) in 3x3 / 1x5 / 5x1 -> split operands for
    gru_conv_zr / gru_conv_q."""
    ks = tuple(weight.shape[2:])
    if ks == (3, 3):
        return conv3x3_split_filter(weight, pieces)
    if ks in ((1, 5), (5, 1)):
        return conv5_split_filter(weight, pieces)
    raise ValueError("GRU gate convolutions are 3x3, 1x5 or 5x1")


def gru_conv_zr(buf: torch.Tensor, arranged: torch.Tensor, bias, cz, cr, h: torch.Tensor, kernel_size) -> tuple:
    """The merged z|r convolution of a GRU step over buf = [h | inputs] with the gate arithmetic in its store:
    returns (z, r*h), each (B,C,H,W).  ``arranged``: gru_split_filter(cat(W_z, W_r)); ``bias`` (2C) or None;
    cz / cr context terms (B,C,H,W slices) or None."""
    B, Cin, H, W = buf.shape
    C = h.shape[1]
    if tuple(h.shape) != (B, C, H, W):
        raise ValueError("gru_conv_zr: h must be (B,C,H,W) matching buf")
    shape = _GRU_SHAPES[tuple(kernel_size)]
    taps = 9 if shape == 0 else 5
    if arranged.numel() != 2 * C * Cin * taps or (bias is not None and bias.numel() != 2 * C):
        raise ValueError("gru_conv_zr: weights / bias do not match (2C, Cin)")
    z = torch.empty((B, C, H, W), device=buf.device, dtype=torch.float32)
    rh = torch.empty((B, C, H, W), device=buf.device, dtype=torch.float32)
    pcz, scz = _slice(cz, "cz", buf); pcr, scr = _slice(cr, "cr", buf); ph, sh = _slice(h, "h", buf)
    _call("vo_gru_conv_zr_bf16x3", _dev(buf, "buf", torch.float32), _dev(arranged, "weights", torch.float32),
          _opt(bias, "bias", torch.float32), pcz, scz, pcr, scr, ph, sh, z.data_ptr(), rh.data_ptr(), B, Cin, H, W, C, shape,
          *_pieces_of(arranged), _stream(), nbytes=4 * (buf.numel() + 3 * z.numel() + z.numel() * ((cz is not None) + (cr is not None))),
          flops=2 * 3 * taps * B * 2 * C * Cin * H * W)
    return z, rh


def gru_conv_q(buf: torch.Tensor, rh: torch.Tensor, arranged: torch.Tensor, bias, cq, z: torch.Tensor, h: torch.Tensor,
               kernel_size, h_copy=None) -> torch.Tensor:
    """The q convolution over [rh | buf[:, C:]] with the state update in its store: returns
    h' = (1-z)*h + z*tanh(conv + bias + cq); ``h_copy`` (a (B,C,H,W) slice, typically buf[:, :C]) receives h' too."""
    B, Cin, H, W = buf.shape
    C = h.shape[1]
    for t, n in ((rh, "rh"), (z, "z"), (h, "h")):
        if tuple(t.shape) != (B, C, H, W):
            raise ValueError(f"gru_conv_q: {n} must be (B,C,H,W) matching buf")
    shape = _GRU_SHAPES[tuple(kernel_size)]
    taps = 9 if shape == 0 else 5
    if arranged.numel() != C * Cin * taps or (bias is not None and bias.numel() != C):
        raise ValueError("gru_conv_q: weights / bias do not match (C, Cin)")
    h_out = torch.empty((B, C, H, W), device=buf.device, dtype=torch.float32)
    pcq, scq = _slice(cq, "cq", buf); ph, sh = _slice(h, "h", buf); pc, sc = _slice(h_copy, "h_copy", buf)
    _call("vo_gru_conv_q_bf16x3", _dev(buf, "buf", torch.float32), _dev(rh, "rh", torch.float32),
          _dev(arranged, "weights", torch.float32), _opt(bias, "bias", torch.float32), pcq, scq, _dev(z, "z", torch.float32),
          ph, sh, h_out.data_ptr(), pc, sc, B, Cin, H, W, C, shape, *_pieces_of(arranged), _stream(),
          nbytes=4 * (buf.numel() + z.numel() * (3 + (cq is not None) + (h_copy is not None))),
          flops=2 * 3 * taps * B * C * Cin * H * W)
    return h_out


# ------------------------------------------------------------------ RAFT ops

def pad_amounts(ht: int, wd: int, divis_by: int):
    """[left, right, top, bottom] replicate padding up to a multiple of ``divis_by``,
    split evenly with the odd pixel at the far side."""
    ph = (((ht // divis_by) + 1) * divis_by - ht) % divis_by
    pw = (((wd // divis_by) + 1) * divis_by - wd) % divis_by
    return [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]


def prepare_images(gray: torch.Tensor, divis_by: int = 8, normalize: bool = True, channels: int = 3):
    """(F,H,W) uint8 GPU frames -> ((F,channels,Hp,Wp) float32, pads): three equal channels (or, with
    ``channels=1``, the one plane they all equal), replicate padding to a multiple of ``divis_by``,
    optional 2*(x/255)-1."""
    if gray.dim() != 3:
        raise ValueError("gray frames must be (F,H,W)")
    if channels not in (1, 3):
        raise ValueError("channels must be 1 or 3")
    F_, H, W = gray.shape
    pads = pad_amounts(H, W, divis_by)
    out = torch.empty((F_, channels, H + pads[2] + pads[3], W + pads[0] + pads[1]), device=gray.device, dtype=torch.float32)
    _call("vo_prepare_images" if channels == 3 else "vo_prepare_images_gray", _dev(gray, "gray", torch.uint8), out.data_ptr(),
          F_, H, W, *pads, int(bool(normalize)), _stream(), nbytes=gray.numel() + 4 * out.numel())
    return out, pads


def pyramid_shapes_2d(h2: int, w2: int, levels: int = MAX_LEVELS):
    out = [(h2, w2)]
    for _ in range(levels - 1):
        out.append((out[-1][0] // 2, out[-1][1] // 2))
    return out


def _split_pieces(split) -> str | None:
    """The correlation functions' ``split`` argument: False -> float32 matrix instructions (None), True or "bf16" ->
    bf16 pieces, "f16" -> float16 pieces."""
    if split is False or split is None:
        return None
    if split is True:
        return "bf16"
    if split in PIECES:
        return split
    raise ValueError(f"split must be False, True, 'bf16' or 'f16', got {split!r}")


def feature_rows_amax(fmap: torch.Tensor) -> torch.Tensor:
    """(B, ..., D) float32 feature maps -> (B,) int32: the float BITS of each image's largest |x|, what the
    float16-piece correlation kernels take their per-image power of two from (it stays on the device)."""
    B = fmap.shape[0]
    amax = torch.empty((B,), device=fmap.device, dtype=torch.int32)
    _call("vo_feature_rows_amax", _dev(fmap, "fmap", torch.float32), amax.data_ptr(), B, fmap.numel() // B, _stream(),
          nbytes=fmap.numel() * 4)
    return amax


def _amax_ptrs(pieces, fmap1, fmap2, amax1, amax2):
    if pieces != "f16":
        return None, None, None, None
    amax1 = feature_rows_amax(fmap1) if amax1 is None else amax1
    amax2 = feature_rows_amax(fmap2) if amax2 is None else amax2
    for a, f in ((amax1, fmap1), (amax2, fmap2)):
        if a.dtype != torch.int32 or a.numel() != f.shape[0] or not a.is_contiguous() or a.device != f.device:
            raise ValueError("amax must be a contiguous int32 tensor with one entry per image (feature_rows_amax)")
    return amax1, amax2, amax1.data_ptr(), amax2.data_ptr()


def corr_volume_2d(fmap1: torch.Tensor, fmap2: torch.Tensor, split=False, amax1=None, amax2=None) -> torch.Tensor:
    """All-pairs correlation of two pixel-major feature maps (B,H,W,D) ->
    (B, H*W, H, W) float32, scaled by 1/sqrt(D).  ``split``: every float32 product from three 16-bit
    matrix-core terms instead of the float32 matrix instruction: True / "bf16" (16 significant bits per
    operand) or "f16" (22 bits; each image's map scaled by a power of two taken from its largest |x|:
    ``amax1`` / ``amax2`` from feature_rows_amax, computed here when absent)."""
    B, H, W, D = fmap1.shape
    if fmap2.shape != fmap1.shape:
        raise ValueError("fmap1 and fmap2 must have the same shape")
    pieces = _split_pieces(split)
    corr = torch.empty((B, H * W, H, W), device=fmap1.device, dtype=torch.float32)
    if pieces is None:
        _call("vo_corr_volume_2d", _dev(fmap1, "fmap1", torch.float32), _dev(fmap2, "fmap2", torch.float32),
              corr.data_ptr(), B, H, W, D, _stream())
        return corr
    a1, a2, p1, p2 = _amax_ptrs(pieces, fmap1, fmap2, amax1, amax2)
    _call("vo_corr_volume_2d_bf16x3", _dev(fmap1, "fmap1", torch.float32), _dev(fmap2, "fmap2", torch.float32),
          corr.data_ptr(), B, H, W, D, PIECES[pieces], p1, p2, _stream())
    return corr


def corr_volume_1d(fmap1: torch.Tensor, fmap2: torch.Tensor, split=False, amax1=None, amax2=None) -> torch.Tensor:
    """Per-row all-pairs correlation (B,H,W1,D) x (B,H,W2,D) -> (B,H,W1,W2); ``split`` as corr_volume_2d."""
    B, H, W1, D = fmap1.shape
    if fmap2.shape[0] != B or fmap2.shape[1] != H or fmap2.shape[3] != D:
        raise ValueError("fmap2 must be (B,H,W2,D) matching fmap1")
    W2 = fmap2.shape[2]
    pieces = _split_pieces(split)
    corr = torch.empty((B, H, W1, W2), device=fmap1.device, dtype=torch.float32)
    if pieces is None:
        _call("vo_corr_volume_1d", _dev(fmap1, "fmap1", torch.float32), _dev(fmap2, "fmap2", torch.float32),
              corr.data_ptr(), B, H, W1, W2, D, _stream())
        return corr
    a1, a2, p1, p2 = _amax_ptrs(pieces, fmap1, fmap2, amax1, amax2)
    _call("vo_corr_volume_1d_bf16x3", _dev(fmap1, "fmap1", torch.float32), _dev(fmap2, "fmap2", torch.float32),
          corr.data_ptr(), B, H, W1, W2, D, PIECES[pieces], p1, p2, _stream())
    return corr


def split_feature_rows(fmap: torch.Tensor, block: int, pieces: str = "bf16", amax: torch.Tensor | None = None) -> torch.Tensor:
    """Pixel-major feature maps (B, ..., D) float32 -> a tensor of the same shape and dtype whose BYTES are the
    maps' 16-bit pieces: per row and block of ``block`` features, ``block`` high parts then ``block`` low
    parts.  "bf16": high = bf16(x), low = bf16(x - high).  "f16": the same on x * 2^e with float16 parts, e per image
    from ``amax`` (feature_rows_amax(fmap): pass the same tensor to the correlation call).  What the ``presplit``
    correlation kernels read: 32 for corr_volume_2d_presplit, 16 for corr_volume_1d_presplit.  The values are not
    float32 numbers any more."""
    D = fmap.shape[-1]
    out = torch.empty_like(fmap)
    rows = fmap.numel() // D
    if pieces == "f16":
        if amax is None or amax.dtype != torch.int32 or amax.numel() != fmap.shape[0] or not amax.is_contiguous():
            raise ValueError("float16 pieces need amax = feature_rows_amax(fmap)")
        _call("vo_split_feature_rows", _dev(fmap, "fmap", torch.float32), out.data_ptr(), rows, D, block, PIECES[pieces],
              amax.data_ptr(), rows // fmap.shape[0], _stream(), nbytes=fmap.numel() * 8)
    else:
        _call("vo_split_feature_rows", _dev(fmap, "fmap", torch.float32), out.data_ptr(), rows, D, block, PIECES[pieces], None, 0, _stream(),
              nbytes=fmap.numel() * 8)
    return out


def corr_volume_2d_presplit(fmap1: torch.Tensor, fmap2: torch.Tensor, pieces: str = "bf16", amax1=None, amax2=None) -> torch.Tensor:
    """corr_volume_2d(split=pieces) on feature maps already in split_feature_rows(., 32, pieces) form; float16
    pieces with the ``amax`` entries of the two maps' images."""
    B, H, W, D = fmap1.shape
    if fmap2.shape != fmap1.shape:
        raise ValueError("fmap1 and fmap2 must have the same shape")
    if pieces == "f16" and (amax1 is None or amax2 is None):
        raise ValueError("float16 pieces need the amax entries the maps were split with")
    a1, a2, p1, p2 = _amax_ptrs(pieces, fmap1, fmap2, amax1, amax2)
    corr = torch.empty((B, H * W, H, W), device=fmap1.device, dtype=torch.float32)
    _call("vo_corr_volume_2d_presplit", _dev(fmap1, "fmap1", torch.float32), _dev(fmap2, "fmap2", torch.float32),
          corr.data_ptr(), B, H, W, D, PIECES[pieces], p1, p2, _stream())
    return corr


def corr_volume_1d_presplit(fmap1: torch.Tensor, fmap2: torch.Tensor, pieces: str = "bf16", amax1=None, amax2=None) -> torch.Tensor:
    """corr_volume_1d(split=pieces) on feature maps already in split_feature_rows(., 16, pieces) form."""
    B, H, W1, D = fmap1.shape
    if fmap2.shape[0] != B or fmap2.shape[1] != H or fmap2.shape[3] != D:
        raise ValueError("fmap2 must be (B,H,W2,D) matching fmap1")
    if pieces == "f16" and (amax1 is None or amax2 is None):
        raise ValueError("float16 pieces need the amax entries the maps were split with")
    W2 = fmap2.shape[2]
    a1, a2, p1, p2 = _amax_ptrs(pieces, fmap1, fmap2, amax1, amax2)
    corr = torch.empty((B, H, W1, W2), device=fmap1.device, dtype=torch.float32)
    _call("vo_corr_volume_1d_presplit", _dev(fmap1, "fmap1", torch.float32), _dev(fmap2, "fmap2", torch.float32),
          corr.data_ptr(), B, H, W1, W2, D, PIECES[pieces], p1, p2, _stream())
    return corr


def corr_pyramid_2d(corr: torch.Tensor, levels: int = MAX_LEVELS):
    """[(B,N,H2,W2) level 0, ...floor-halved levels] from the level-0 volume."""
    if not 1 <= levels <= MAX_LEVELS:
        raise ValueError("levels must be in 1..4")
    B, N, H2, W2 = corr.shape
    shapes = pyramid_shapes_2d(H2, W2, levels)
    pyr = [corr] + [torch.empty((B, N, h, w), device=corr.device, dtype=torch.float32) for h, w in shapes[1:]]
    ptrs = [p.data_ptr() for p in pyr[1:]] + [0] * (MAX_LEVELS - levels)
    if levels > 1:
        _call("vo_corr_pyramid_2d", _dev(corr, "corr", torch.float32), *ptrs, B * N, H2, W2, _stream())
    return pyr


def corr_pyramid_1d(corr: torch.Tensor, levels: int = MAX_LEVELS):
    """[(B,H,W1,W2), (B,H,W1,W2//2), ...] from the level-0 row volume."""
    if not 1 <= levels <= MAX_LEVELS:
        raise ValueError("levels must be in 1..4")
    B, H, W1, W2 = corr.shape
    widths = [W2]
    for _ in range(levels - 1):
        widths.append(widths[-1] // 2)
    pyr = [corr] + [torch.empty((B, H, W1, w), device=corr.device, dtype=torch.float32) for w in widths[1:]]
    ptrs = [p.data_ptr() for p in pyr[1:]] + [0] * (MAX_LEVELS - levels)
    if levels > 1:
        _call("vo_corr_pyramid_1d", _dev(corr, "corr", torch.float32), *ptrs, B * H * W1, W2, _stream())
    return pyr


def _level_ptrs(pyr):
    if not 1 <= len(pyr) <= MAX_LEVELS:
        raise ValueError("pyramid must have 1..4 levels")
    return [_dev(p, f"pyr[{i}]", torch.float32) for i, p in enumerate(pyr)] + [0] * (MAX_LEVELS - len(pyr))


def corr_lookup_2d(pyr, coords: torch.Tensor, radius: int = 4, out: torch.Tensor | None = None) -> torch.Tensor:
    """Windowed bilinear lookup -> (B, L*(2r+1)^2, H, W)."""
    B, two, H, W = coords.shape
    _, N, H2, W2 = pyr[0].shape
    if two != 2 or N != H * W or pyr[0].shape[0] != B:
        raise ValueError("coords must be (B,2,H,W) with H*W source pixels matching the pyramid")
    for i, (h, w) in enumerate(pyramid_shapes_2d(H2, W2, len(pyr))):
        if tuple(pyr[i].shape) != (B, N, h, w):
            raise ValueError(f"pyramid level {i} has shape {tuple(pyr[i].shape)}, expected {(B, N, h, w)}")
    win = 2 * radius + 1
    ch = len(pyr) * win * win
    if out is None:
        out = torch.empty((B, ch, H, W), device=coords.device, dtype=torch.float32)
    elif tuple(out.shape) != (B, ch, H, W) or out.dtype != torch.float32:
        raise ValueError("out has the wrong shape")
    po, so = _slice(out, "out")                          # may be the leading channels of a wider tensor
    _call("vo_corr_lookup_2d", *_level_ptrs(pyr), _dev(coords, "coords", torch.float32), po, so, B, H, W, H2, W2, len(pyr),
          radius, _stream())
    return out


def corr_lookup_1d(pyr, coords: torch.Tensor, radius: int = 4, out: torch.Tensor | None = None) -> torch.Tensor:
    """Windowed linear lookup along x -> (B, L*(2r+1), H, W); reads coords[:,0]."""
    B, two, H, W = coords.shape
    if two != 2 or tuple(pyr[0].shape[:3]) != (B, H, W):
        raise ValueError("coords must be (B,2,H,W) matching the (B,H,W1,W2) pyramid")
    W2 = pyr[0].shape[3]
    w = W2
    for i, p in enumerate(pyr):
        if tuple(p.shape) != (B, H, W, w):
            raise ValueError(f"pyramid level {i} has shape {tuple(p.shape)}, expected {(B, H, W, w)}")
        w //= 2
    ch = len(pyr) * (2 * radius + 1)
    if out is None:
        out = torch.empty((B, ch, H, W), device=coords.device, dtype=torch.float32)
    elif tuple(out.shape) != (B, ch, H, W) or out.dtype != torch.float32:
        raise ValueError("out has the wrong shape")
    po, so = _slice(out, "out")
    _call("vo_corr_lookup_1d", *_level_ptrs(pyr), _dev(coords, "coords", torch.float32), po, so, B, H, W, W2, len(pyr), radius,
          _stream())
    return out


def convex_upsample(flow: torch.Tensor, mask: torch.Tensor, factor: int, out_channels: int | None = None,
                    mask_scale: float = 1.0) -> torch.Tensor:
    """Convex-combination upsampling -> (B, Cout, f*H, f*W)."""
    B, Cf, H, W = flow.shape
    if tuple(mask.shape) != (B, 9 * factor * factor, H, W):
        raise ValueError(f"mask must be (B,{9 * factor * factor},H,W)")
    Cout = Cf if out_channels is None else out_channels
    if not 1 <= Cout <= Cf:
        raise ValueError("out_channels must be in 1..flow channels")
    out = torch.empty((B, Cout, factor * H, factor * W), device=flow.device, dtype=torch.float32)
    _call("vo_convex_upsample", _dev(flow, "flow", torch.float32), _dev(mask, "mask", torch.float32),
              out.data_ptr(), B, Cf, Cout, H, W, factor, float(mask_scale), _stream())
    return out


# -------------------------------------------------------------- geometry ops

def disparity_to_depth(disparity: torch.Tensor, fxB: float, divide_in_f64: bool = True) -> torch.Tensor:
    depth = torch.empty_like(disparity)
    _call("vo_disparity_to_depth", _dev(disparity, "disparity", torch.float32), depth.data_ptr(),
              disparity.numel(), float(fxB), int(bool(divide_in_f64)), _stream())
    return depth


def _kp_args(kp: torch.Tensor, batch: int):
    """(N,2) shared grid or (B,N,2) per-item keypoints, float64 -> (ptr, stride, N)."""
    p = _dev(kp, "keypoints", torch.float64)
    if kp.dim() == 2 and kp.shape[1] == 2:
        return p, 0, kp.shape[0]
    if kp.dim() == 3 and kp.shape[0] == batch and kp.shape[2] == 2:
        return p, kp.shape[1] * 2, kp.shape[1]
    raise ValueError("keypoints must be (N,2) or (B,N,2)")


def keypoints_to_3d(kp: torch.Tensor, depth: torch.Tensor, K_inv) -> torch.Tensor:
    """(B,H,W) depth + keypoints -> (B,N,3) float64."""
    if depth.dim() != 3:
        raise ValueError("depth must be (B,H,W)")
    B, H, W = depth.shape
    p, stride, N = _kp_args(kp, B)
    out = torch.empty((B, N, 3), device=depth.device, dtype=torch.float64)
    _call("vo_keypoints_to_3d", p, stride, _dev(depth, "depth", torch.float32), _host9(K_inv),
              out.data_ptr(), B, N, H, W, _stream())
    return out


def keypoints_flow_2d(kp: torch.Tensor, flow: torch.Tensor):
    """(B,2,H,W) flow -> next (B,N,2) float64, valid (B,N) bool."""
    if flow.dim() != 4 or flow.shape[1] != 2:
        raise ValueError("flow must be (B,2,H,W)")
    B, _, H, W = flow.shape
    p, stride, N = _kp_args(kp, B)
    nxt = torch.empty((B, N, 2), device=flow.device, dtype=torch.float64)
    valid = torch.empty((B, N), device=flow.device, dtype=torch.uint8)
    _call("vo_keypoints_flow_2d", p, stride, _dev(flow, "flow", torch.float32), nxt.data_ptr(),
              valid.data_ptr(), B, N, H, W, _stream())
    return nxt, valid.bool()


def _check_maps(depth1, depth2, flow):
    if flow.dim() != 4 or flow.shape[1] != 2:
        raise ValueError("flow must be (B,2,H,W)")
    B, _, H, W = flow.shape
    if tuple(depth1.shape) != (B, H, W) or tuple(depth2.shape) != (B, H, W):
        raise ValueError("depth maps must be (B,H,W) matching the flow")
    return B, H, W


def keypoints_flow_3d(kp, depth1, depth2, flow, K_inv, want_first: bool = False):
    B, H, W = _check_maps(depth1, depth2, flow)
    p, stride, N = _kp_args(kp, B)
    p2 = torch.empty((B, N, 3), device=flow.device, dtype=torch.float64)
    p1 = torch.empty_like(p2) if want_first else None
    valid = torch.empty((B, N), device=flow.device, dtype=torch.uint8)
    _call("vo_keypoints_flow_3d", p, stride, _dev(depth1, "depth1", torch.float32),
              _dev(depth2, "depth2", torch.float32), _dev(flow, "flow", torch.float32), _host9(K_inv),
              p2.data_ptr(), valid.data_ptr(), 0 if p1 is None else p1.data_ptr(), B, N, H, W, _stream())
    return (p2, valid.bool(), p1) if want_first else (p2, valid.bool())


def pair_correspondences(kp, depth1, depth2, flow, K_inv, min_depth=0.0, max_depth=15.0, max_dz=1.0):
    """Per-pair weighted correspondences -> P1, P2 (B,N,3) f64, weight (B,N) uint8,
    n_in_range (B) int32."""
    B, H, W = _check_maps(depth1, depth2, flow)
    p, stride, N = _kp_args(kp, B)
    dev = flow.device
    p1 = torch.empty((B, N, 3), device=dev, dtype=torch.float64)
    p2 = torch.empty_like(p1)
    wgt = torch.empty((B, N), device=dev, dtype=torch.uint8)
    n_in = torch.empty((B,), device=dev, dtype=torch.int32)
    _call("vo_pair_correspondences", p, stride, _dev(depth1, "depth1", torch.float32),
              _dev(depth2, "depth2", torch.float32), _dev(flow, "flow", torch.float32), _host9(K_inv),
              float(min_depth), float(max_depth), float(max_dz), p1.data_ptr(), p2.data_ptr(),
              wgt.data_ptr(), n_in.data_ptr(), B, N, H, W, _stream())
    return p1, p2, wgt, n_in


def pair_poses(kp, depth1, depth2, flow, K_inv, min_depth=0.0, max_depth=15.0, max_dz=1.0, offset=None,
               min_points: int = 4, workspace: torch.Tensor | None = None):
    """The geometry half of the tracking loop in one call: grid keypoints + depth maps (B,H,W) + flow
    (B,2,H,W) -> T (B,4,4) float64, skipped (B) bool.  ``workspace`` (optional uint8 GPU tensor,
    reused across calls when it is large enough) holds the correspondences; returns it third."""
    B, H, W = _check_maps(depth1, depth2, flow)
    p, stride, N = _kp_args(kp, B)
    if N < 1:
        raise ValueError("pair_poses needs at least one keypoint")
    need = int(_lib.load().vo_pair_poses_workspace_bytes(B, N))
    if workspace is None or workspace.numel() < need or workspace.device != flow.device:
        workspace = torch.empty((need,), device=flow.device, dtype=torch.uint8)
    T = torch.empty((B, 4, 4), device=flow.device, dtype=torch.float64)
    skipped = torch.empty((B,), device=flow.device, dtype=torch.uint8)
    off = None
    if offset is not None:
        off = _host_doubles(offset, 3, "offset")
    _call("vo_pair_poses", p, stride, _dev(depth1, "depth1", torch.float32), _dev(depth2, "depth2", torch.float32),
          _dev(flow, "flow", torch.float32), _host9(K_inv), float(min_depth), float(max_depth), float(max_dz), off,
          int(min_points), T.data_ptr(), skipped.data_ptr(), _dev(workspace, "workspace", torch.uint8),
          workspace.numel(), B, N, H, W, _stream(), nbytes=B * N * (16 + 8 + 8 + 2 * 49) + B * 129)
    return T, skipped.bool(), workspace


def keypoints_to_2d(points: torch.Tensor, K):
    """(..., 3) float64 camera-space points -> (uv (..., 2) float64, keep (...) bool); keep is
    False where Z <= 0 (rows the reference's to_2d drops)."""
    if points.shape[-1] != 3:
        raise ValueError("points must be (..., 3)")
    n = points.numel() // 3
    uv = torch.empty(points.shape[:-1] + (2,), device=points.device, dtype=torch.float64)
    keep = torch.empty(points.shape[:-1], device=points.device, dtype=torch.uint8)
    _call("vo_keypoints_to_2d", _dev(points, "points", torch.float64), _host9(K), uv.data_ptr(), keep.data_ptr(),
          n, _stream())
    return uv, keep.bool()


def track_keypoints(grid_kp: torch.Tensor, depth: torch.Tensor, flow: torch.Tensor, K, K_inv, k: int = 25,
                    min_points: int = 4, first_index: int = 0):
    """Multi-frame keypoint tracks over depth (F,H,W) and flow (F-1,2,H,W) ->
    tracks (F-1,N,3) float64 (x, y, Z), alive (F-1,N) bool, reseeded (F-1) bool."""
    if depth.dim() != 3:
        raise ValueError("depth must be (F,H,W)")
    F, H, W = depth.shape
    if F < 1 or tuple(flow.shape) != (F - 1, 2, H, W):
        raise ValueError("flow must be (F-1,2,H,W) matching the depth maps")
    if grid_kp.dim() != 2 or grid_kp.shape[1] != 2:
        raise ValueError("grid keypoints must be (N,2)")
    if k < 1 or first_index < 0 or min_points < 0:
        raise ValueError("k >= 1, first_index >= 0 and min_points >= 0 required")
    N = grid_kp.shape[0]
    dev = depth.device
    tracks = torch.empty((F - 1, N, 3), device=dev, dtype=torch.float64)
    alive = torch.empty((F - 1, N), device=dev, dtype=torch.uint8)
    reseeded = torch.empty((F - 1,), device=dev, dtype=torch.uint8)
    if F > 1:
        _call("vo_track_keypoints", _dev(grid_kp, "keypoints", torch.float64) if N else 0,
              _dev(depth, "depth", torch.float32), _dev(flow, "flow", torch.float32), _host9(K), _host9(K_inv),
              int(first_index), int(k), int(min_points), tracks.data_ptr(), alive.data_ptr(), reseeded.data_ptr(),
              F, N, H, W, _stream())
    return tracks, alive.bool(), reseeded.bool()


def kabsch_batch(P: torch.Tensor, Q: torch.Tensor, weight: torch.Tensor | None = None, offset=None):
    """(B,N,3) float64 point sets -> T (B,4,4) float64, n_used (B) int32."""
    if P.dim() != 3 or P.shape[2] != 3 or Q.shape != P.shape:
        raise ValueError("P and Q must both be (B,N,3)")
    B, N, _ = P.shape
    if weight is not None and tuple(weight.shape) != (B, N):
        raise ValueError("weight must be (B,N)")
    T = torch.empty((B, 4, 4), device=P.device, dtype=torch.float64)
    n_used = torch.empty((B,), device=P.device, dtype=torch.int32)
    off = None
    if offset is not None:
        o = np.asarray(offset, dtype=np.float64).reshape(-1)
        if o.size != 3:
            raise ValueError("offset must have 3 elements")
        off = (ctypes.c_double * 3)(*o.tolist())
    _call("vo_kabsch_batch", _dev(P, "P", torch.float64), _dev(Q, "Q", torch.float64),
              _opt(weight, "weight", torch.uint8), off, T.data_ptr(), n_used.data_ptr(), B, N, _stream())
    return T, n_used


def ransac_kabsch_batch(P: torch.Tensor, Q: torch.Tensor, weight: torch.Tensor | None = None,
                        threshold: float = 0.05, iterations: int = 1000, seed: int = 0, want_mask: bool = False,
                        first_pair: int = 0):
    """(B,N,3) float64 point sets -> T (B,4,4) f64, n_inliers (B) int32 [, inlier mask (B,N) bool].
    ``first_pair`` is item 0's pair index in the whole sequence: the hypothesis triples are keyed
    on that global index, so chunking or sharding a sequence does not change its poses."""
    if P.dim() != 3 or P.shape[2] != 3 or Q.shape != P.shape:
        raise ValueError("Input arrays must be of shape (N, 3)")
    B, N, _ = P.shape
    if N < 3:
        raise ValueError("At least 3 points are required for RANSAC.")
    if weight is not None and tuple(weight.shape) != (B, N):
        raise ValueError("weight must be (B,N)")
    T = torch.empty((B, 4, 4), device=P.device, dtype=torch.float64)
    n_inl = torch.empty((B,), device=P.device, dtype=torch.int32)
    mask = torch.empty((B, N), device=P.device, dtype=torch.uint8) if want_mask else None
    _call("vo_ransac_kabsch_batch", _dev(P, "P", torch.float64), _dev(Q, "Q", torch.float64),
          _opt(weight, "weight", torch.uint8), float(threshold), int(iterations), int(seed) & (2 ** 64 - 1),
          T.data_ptr(), n_inl.data_ptr(), 0 if mask is None else mask.data_ptr(), B, N, int(first_pair), _stream())
    return (T, n_inl, mask.bool()) if want_mask else (T, n_inl)
