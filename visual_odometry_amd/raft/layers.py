"""Convolutional encoders of RAFT and RAFT-Stereo.

The encoders are ``nn.Module`` trees that hold the parameters under the published models' names
(``conv1``, ``norm1``, ``layer1``.., ``conv2``, ``outputs08``..), so that a state dict trained with those
models, ``module.`` prefix stripped, loads unchanged; the reference builds them with ``RAFT(args)`` /
``RAFTStereo(args)`` at flow/flow_map_RAFT.py:42 and stereo/stereo_disparity_RAFT.py:46.  Inside a
product forward pass (``fused_passes()``) their arithmetic does NOT go to the framework's operators: every
convolution of the float32 path is dispatched to a hand-written matrix-core kernel of ``ops`` (3x3, 1x5 /
5x1, stride-2, 7x7 and 1x1 kernels with bias / folded batch norm / ReLU / residual join in their stores),
instance norms are applied by the next kernel that reads the tensor, and the remaining passes are fused
in-place HIP kernels.  With ``fused_passes()`` off the same modules run the stock PyTorch operators.

Which kernels a forward pass uses is decided by the ``Arithmetic`` in force (``arithmetic(...)``), a
context-local value that each model sets from its own ``conv_arithmetic`` field for the duration of its
forward pass -- two pipelines in one process cannot change each other's arithmetic.
"""
from __future__ import annotations

import contextlib
import contextvars
import dataclasses

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops

CONV_ARITHMETICS = ("f16x3", "bf16x3", "f32")


@dataclasses.dataclass(frozen=True)
class Arithmetic:
    """How a forward pass computes.
      kind           how the heavy float32 products (3x3, 1x5 / 5x1, stride-2, many-channel 1x1 and 7x7 convolutions,
                     GRU gate convolutions, correlation volumes) are formed:
                       "f16x3"   three float16 matrix-core terms per product, x*w ~ xl*wh + xh*wl + xh*wh with both
                                 operands cut into two float16 pieces: 22 significant bits per operand, float32
                                 accumulation; the error of a layer is that of the float32 matrix instruction.  The
                                 correlation volumes likewise, every image's feature map scaled by the power of two
                                 of its largest |x| before it is cut (found on the device, per image).
                       "bf16x3"  the same with bf16 pieces: 16 significant bits per operand, largest error 4-6e-6 of a
                                 layer's largest output; correlation volumes included.
                       "f32"     float32 matrix instructions only (Winograd kernels for 3x3 and 1x5 / 5x1; the
                                 F(4x4,3x3) form's rounding error is 1e-5 of the largest output).
      fused          norm -> ReLU, residual joins and GRU gates as fused HIP passes and every convolution on the
                     hand-written kernels (the product's forward passes switch it on); off: stock PyTorch operators.
      hand_kernels   off: the convolutions themselves go to the framework (a switch for comparisons).
      winograd_tile  0: F(2x2,3x3) or F(4x4,3x3) per layer shape in the "f32" path; 2 or 4 forces one."""
    kind: str = "f16x3"
    fused: bool = False
    hand_kernels: bool = True
    winograd_tile: int = 0


_CURRENT = contextvars.ContextVar("vo_conv_arithmetic", default=Arithmetic())


def current() -> Arithmetic:
    return _CURRENT.get()


@contextlib.contextmanager
def arithmetic(kind=None, *, fused=None, hand_kernels=None, winograd_tile=None):
    """Put an Arithmetic in force for the enclosed calls (of this thread / context); fields left at None keep
    the value of the enclosing scope."""
    changes = {k: v for k, v in dict(kind=kind, fused=fused, hand_kernels=hand_kernels, winograd_tile=winograd_tile).items()
               if v is not None}
    if "kind" in changes and changes["kind"] not in CONV_ARITHMETICS:
        raise ValueError(f"conv arithmetic must be one of {CONV_ARITHMETICS}")
    if "winograd_tile" in changes and changes["winograd_tile"] not in (0, 2, 4):
        raise ValueError("winograd_tile must be 0, 2 or 4")
    token = _CURRENT.set(dataclasses.replace(_CURRENT.get(), **changes))
    try:
        yield
    finally:
        _CURRENT.reset(token)


def fused_passes(on: bool = True):
    """The product's forward passes: fused HIP passes and hand-written convolution kernels on."""
    return arithmetic(fused=bool(on))


def fused_on() -> bool:
    return _CURRENT.get().fused and not torch.is_autocast_enabled()


def split_on() -> bool:
    """True when the heavy float32 products are assembled from three 16-bit matrix-core terms."""
    return _CURRENT.get().kind in ("f16x3", "bf16x3")


def split_pieces() -> str:
    """The piece format of the split kernels in force: "f16" or "bf16"."""
    return "f16" if _CURRENT.get().kind == "f16x3" else "bf16"


def bf16_pieces_on() -> bool:
    """True under "bf16x3": the kernels that exist for bf16 pieces only (correlation volumes, rows split by
    their producer, the weights-in-LDS 1x1 kernel) apply."""
    return _CURRENT.get().kind == "bf16x3"


_ACT_DTYPES = (torch.float32, torch.float16)


def _fusable(x: torch.Tensor) -> bool:
    return fused_on() and x.is_cuda and x.dtype in _ACT_DTYPES and x.is_contiguous()


def weight_for(conv: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """The convolution's weight in the dtype of its input: the float32 parameter itself,
    or a float16 copy cached on the module (rebuilt when the parameter changes or moves).
    The mixed-precision path keeps float32 master weights and float16 activations."""
    if x.dtype == conv.weight.dtype:
        return conv.weight
    key = (conv.weight._version, conv.weight.data_ptr(), x.dtype)
    cached = getattr(conv, "_vo_weight_cast", None)
    if cached is None or cached[0] != key:
        cached = (key, conv.weight.detach().to(x.dtype).contiguous(),
                  None if conv.bias is None else conv.bias.detach().to(x.dtype).contiguous())
        conv._vo_weight_cast = cached
    return cached[1]


def conv_plain(x: torch.Tensor, conv: nn.Conv2d) -> torch.Tensor:
    """conv(x) with bias, weights cast to the activations' dtype when they differ."""
    if x.dtype == conv.weight.dtype:
        if fused_on() and conv.bias is not None and x.dtype == torch.float32 and x.is_cuda:
            U = _winograd_filter_for(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
            if U is not None:
                return ops.conv3x3_winograd(x, U, conv.out_channels, None, conv.bias, False)
            Ws = _conv1x1_split_weights_for(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
            if Ws is not None:                                   # many channels on both sides (the mask heads' last layer)
                return ops.conv1x1_split(x, Ws, conv.out_channels, None, conv.bias, False)
            Wg = _conv1x1_weights_for(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
            if Wg is not None:
                return ops.conv1x1(x, Wg, conv.out_channels, None, conv.bias, False)
            Wg = _conv7x7_weights_for(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
            if Wg is not None:
                return ops.conv7x7(x, Wg, conv.out_channels, None, conv.bias, False, conv.stride[0])
            Us = _split_s2_weights_for(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
            if Us is not None:
                return ops.conv3x3s2_split(x, Us, conv.out_channels, None, conv.bias, False)
            Wg = _conv3x3s2_weights_for(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
            if Wg is not None:
                return ops.conv3x3s2(x, Wg, conv.out_channels, None, conv.bias, False)
            if (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
                    and conv.groups == 1 and conv.out_channels <= 4 and x.is_contiguous()):
                # very few output channels (the flow heads): a 1x1 convolution with one output plane
                # per filter tap (a skinny GEMM that reads x once), then one pass that adds the nine
                # shifted planes -- MIOpen's 3x3 kernels cost as much here as for 64 output channels
                key = (conv.weight._version, conv.weight.data_ptr())
                cached = getattr(conv, "_vo_taps", None)
                if cached is None or cached[0] != key:
                    K, C = conv.weight.shape[:2]
                    cached = (key, conv.weight.detach().permute(0, 2, 3, 1).reshape(K * 9, C, 1, 1).contiguous())
                    conv._vo_taps = cached
                taps = cached[1]
                Wg = _conv1x1_weights_for(x, taps, (1, 1), (0, 0), (1, 1), 1)
                planes = ops.conv1x1(x, Wg, taps.shape[0]) if Wg is not None else F.conv2d(x, taps)
                return ops.shift_add_3x3(planes, conv.bias, conv.out_channels)
        return conv(x)
    w = weight_for(conv, x)
    return F.conv2d(x, w, conv._vo_weight_cast[2], conv.stride, conv.padding, conv.dilation, conv.groups)


def conv_pixel_major(x: torch.Tensor, conv: nn.Conv2d, split_rows: bool = False) -> torch.Tensor:
    """conv(x) with bias as float32 (N,H,W,K): each pixel's feature vector contiguous, the layout
    the correlation kernels read.  A float32 1x1 convolution writes it directly; anything else is
    the convolution followed by the permuting copy.  ``split_rows``: the rows as bf16 pieces in blocks
    of 32 channels (ops.split_feature_rows(., 32)), written by the convolution's store where that
    kernel applies, by a pass over the float32 rows otherwise."""
    if (x.dtype == torch.float32 and conv.weight.dtype == torch.float32 and fused_on() and conv.bias is not None and x.is_cuda):
        if conv.out_channels % (32 if split_rows else 4) == 0 and x.is_contiguous():
            Ws = _conv1x1_split_weights_for(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
            if Ws is not None:
                return ops.conv1x1_pixel_major_split(x, Ws, conv.out_channels, None, conv.bias, False, split_rows)
        Wg = _conv1x1_weights_for(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
        if Wg is not None:
            y = ops.conv1x1_pixel_major(x, Wg, conv.out_channels, None, conv.bias, False)
            return ops.split_feature_rows(y, 32) if split_rows else y
    y = conv_plain(x, conv).float().permute(0, 2, 3, 1).contiguous()
    return ops.split_feature_rows(y, 32) if split_rows else y


def _hand() -> bool:
    return _CURRENT.get().hand_kernels


def _split_cached(weight: torch.Tensor, attr: str, make):
    """The split operands of ``weight`` for the piece format in force, cached on the weight tensor per
    format; rebuilt when the weight changes or moves."""
    pieces = split_pieces()
    key = (weight._version, weight.data_ptr())
    table = getattr(weight, attr, None)
    if table is None or table[0] != key:
        table = (key, {})
        setattr(weight, attr, table)
    A = table[1].get(pieces)
    if A is None:
        A = table[1][pieces] = make(weight, pieces)
    return A


def _winograd_tile_for(H: int, W: int, C: int = 0) -> int:
    """Which 3x3 kernel: 1 = the split-bf16 direct kernel, 4 / 2 = Winograd F(4x4,3x3) / F(2x2,3x3).
    F(4x4,3x3) does a pixel in about 1/1.55 of F(2x2,3x3)'s time but works on 16x32- or 32x16-pixel
    blocks (whichever covers the map with less padding) instead of 16x16: take it unless the padding
    of small maps eats the gain.  The split-bf16 kernel does a pixel in about 1/1.3 of F(4x4,3x3)'s
    time on 16x32-pixel blocks; it needs input channels in multiples of 16 and an even width."""
    if _CURRENT.get().winograd_tile:
        return _CURRENT.get().winograd_tile
    up = lambda v, m: (v + m - 1) // m * m
    covered4 = min(up(H, 16) * up(W, 32), up(H, 32) * up(W, 16))
    best = (4, covered4) if covered4 < 1.55 * up(H, 16) * up(W, 16) else (2, 1.55 * up(H, 16) * up(W, 16))
    if split_on() and C % 16 == 0 and C > 0 and W % 2 == 0 and min(up(H, 16) * up(W, 32), up(H, 32) * up(W, 16)) < 1.3 * best[1]:
        return 1
    return best[0]


def _winograd_filter_for(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups):
    """The cached transformed filter when this convolution can run on a Winograd matrix-core
    kernel (3x3 / stride 1 / padding 1, float32, >= 32 output channels, input channels a
    multiple of 8), else None."""
    if not (_hand() and x.dtype == torch.float32 and x.is_cuda and x.is_contiguous() and weight.dtype == torch.float32
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1)
            and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] % 8 == 0 and weight.shape[0] >= 32
            and weight.shape[1] == x.shape[1]):
        return None
    # the transformed filters ride on the weight tensor object itself (an address-keyed
    # table would hand a new model the filters of a freed one), one per kernel variant
    tile = _winograd_tile_for(x.shape[2], x.shape[3], x.shape[1])
    if tile == 1:
        return _split_cached(weight, "_vo_conv3x3_split", lambda w, pieces: ops.conv3x3_split_filter(w.contiguous(), pieces))
    key = (weight._version, weight.data_ptr())
    table = getattr(weight, "_vo_winograd", None)
    if table is None or table[0] != key:
        table = (key, {})
        weight._vo_winograd = table
    U = table[1].get(tile)
    if U is None:
        U = table[1][tile] = ops.winograd_filter(weight.contiguous(), tile)
    return U


def _conv1x1_weights_for(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups):
    """The cached arranged weights when this convolution can run on the streaming 1x1 matrix-core
    kernel (1x1 / stride 1 / no padding, float32, contiguous input), else None."""
    if not (_hand() and x.dtype == torch.float32 and x.is_cuda and x.is_contiguous() and weight.dtype == torch.float32
            and tuple(weight.shape[2:]) == (1, 1) and tuple(stride) == (1, 1) and tuple(padding) == (0, 0)
            and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] == x.shape[1]
            and (x.shape[1] + 8) * x.shape[2] * x.shape[3] * 4 < 2 ** 31):
        return None
    key = (weight._version, weight.data_ptr())
    cached = getattr(weight, "_vo_conv1x1", None)
    if cached is None or cached[0] != key:
        cached = (key, ops.conv1x1_filter(weight))
        weight._vo_conv1x1 = cached
    return cached[1]


def _conv1x1_split_weights_for(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups):
    """The cached split-bf16 weights when this 1x1 convolution has enough channels on both sides for the matrix
    instruction, not memory, to bound the float32 streaming kernel (>= 64 in, >= 64 out, input channels a
    multiple of 16, even width), else None."""
    if not (_hand() and split_on() and x.dtype == torch.float32 and x.is_cuda and x.is_contiguous()
            and weight.dtype == torch.float32 and tuple(weight.shape[2:]) == (1, 1) and tuple(stride) == (1, 1)
            and tuple(padding) == (0, 0) and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] == x.shape[1]
            and x.shape[1] % 16 == 0 and x.shape[1] >= 64 and weight.shape[0] >= 64 and x.shape[3] % 2 == 0
            and x.shape[1] * x.shape[2] * x.shape[3] * 4 < 2 ** 31 - 1 and weight.shape[0] * x.shape[2] * x.shape[3] * 4 < 2 ** 31 - 1):
        return None
    return _split_cached(weight, "_vo_conv1x1_split", ops.conv1x1_split_filter)


def _conv7x7_weights_for(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups):
    """The cached arranged weights when this convolution can run on the streaming 7x7 matrix-core
    kernel (7x7 / stride 1 or 2 / padding 3, at most 8 input channels, float32), else None."""
    if not (_hand() and x.dtype == torch.float32 and x.is_cuda and x.is_contiguous() and weight.dtype == torch.float32
            and tuple(weight.shape[2:]) == (7, 7) and tuple(stride) in ((1, 1), (2, 2)) and tuple(padding) == (3, 3)
            and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] == x.shape[1] and x.shape[1] <= 8
            and x.shape[1] * x.shape[2] * x.shape[3] * 4 < 2 ** 30):
        return None
    key = (weight._version, weight.data_ptr())
    cached = getattr(weight, "_vo_conv7x7", None)
    if cached is None or cached[0] != key:
        cached = (key, ops.conv7x7_filter(weight))
        weight._vo_conv7x7 = cached
    return cached[1]


def _conv7x7_split_weights_for(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups):
    """The cached split-bf16 weights when this convolution can run on the split-bf16 7x7 kernel (7x7 / stride 1
    or 2 / padding 3, one or two input channels, output channels a multiple of 8, float32), else None."""
    if not (_hand() and split_on() and x.dtype == torch.float32 and x.is_cuda and x.is_contiguous()
            and weight.dtype == torch.float32 and tuple(weight.shape[2:]) == (7, 7) and tuple(stride) in ((1, 1), (2, 2))
            and tuple(padding) == (3, 3) and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] == x.shape[1]
            and x.shape[1] in (1, 2) and weight.shape[0] % 8 == 0 and x.shape[1] * x.shape[2] * x.shape[3] * 4 < 2 ** 30
            and weight.shape[0] * x.shape[2] * x.shape[3] * 4 < 2 ** 31 - 1):
        return None
    return _split_cached(weight, "_vo_conv7x7_split", ops.conv7x7_split_filter)


def _conv1x1s2_split_weights_for(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups):
    """The cached split-bf16 weights when this 1x1 stride-2 convolution (a residual stage's projection
    shortcut) can run on the split-bf16 direct kernel, else None."""
    if not (_hand() and split_on() and x.dtype == torch.float32 and x.is_cuda and x.is_contiguous()
            and weight.dtype == torch.float32 and tuple(weight.shape[2:]) == (1, 1) and tuple(stride) == (2, 2)
            and tuple(padding) == (0, 0) and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] == x.shape[1]
            and x.shape[1] % 16 == 0 and x.shape[3] % 2 == 0 and x.shape[1] * x.shape[2] * x.shape[3] * 4 < 2 ** 31 - 1):
        return None
    wo = (x.shape[3] - 1) // 2 + 1
    if wo < 0.6 * ((wo + 31) // 32 * 32):
        return None
    return _split_cached(weight, "_vo_conv1x1_split", ops.conv1x1_split_filter)


def _conv3x3s2_weights_for(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups):
    """The cached arranged weights when this convolution can run on the streaming 3x3 stride-2
    matrix-core kernel (float32, input channels a multiple of 8, an output width that fills the
    kernel's 32- or 64-pixel row segments to at least 60 %), else None."""
    if not (_hand() and x.dtype == torch.float32 and x.is_cuda and x.is_contiguous() and weight.dtype == torch.float32
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (2, 2) and tuple(padding) == (1, 1)
            and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] == x.shape[1] and x.shape[1] % 8 == 0
            and x.shape[1] * x.shape[2] * x.shape[3] * 4 < 2 ** 30):
        return None
    wo = (x.shape[3] - 1) // 2 + 1
    if wo < 0.6 * ((wo + 31) // 32 * 32):
        return None
    key = (weight._version, weight.data_ptr())
    cached = getattr(weight, "_vo_conv3x3s2", None)
    if cached is None or cached[0] != key:
        cached = (key, ops.conv3x3s2_filter(weight))
        weight._vo_conv3x3s2 = cached
    return cached[1]


def _split_s2_weights_for(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups):
    """The cached split-bf16 weights when this 3x3 stride-2 convolution can run on the split-bf16 direct
    kernel (float32, input channels a multiple of 16, even input width, an output width that fills
    the kernel's 32-pixel row segments to at least 60 %), else None."""
    if not (_hand() and split_on() and x.dtype == torch.float32 and x.is_cuda and x.is_contiguous()
            and weight.dtype == torch.float32 and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (2, 2)
            and tuple(padding) == (1, 1) and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] == x.shape[1]
            and x.shape[1] % 16 == 0 and x.shape[3] % 2 == 0 and x.shape[1] * x.shape[2] * x.shape[3] * 4 < 2 ** 31 - 1):
        return None
    wo = (x.shape[3] - 1) // 2 + 1
    if wo < 0.6 * ((wo + 31) // 32 * 32):
        return None
    # the split filter of a 3x3 weight is shared with stride 1
    return _split_cached(weight, "_vo_conv3x3_split", lambda w, pieces: ops.conv3x3_split_filter(w.contiguous(), pieces))


def conv2d_nobias(x: torch.Tensor, weight: torch.Tensor, stride=(1, 1), padding=(0, 0), dilation=(1, 1),
                  groups: int = 1) -> torch.Tensor:
    """Bias-free convolution inside the fused path: eligible 3x3 layers on the Winograd
    kernel, everything else on MIOpen."""
    U = _winograd_filter_for(x, weight, stride, padding, dilation, groups)
    if U is not None:
        return ops.conv3x3_winograd(x, U, weight.shape[0])
    Ws = _conv1x1_split_weights_for(x, weight, stride, padding, dilation, groups)
    if Ws is not None:
        return ops.conv1x1_split(x, Ws, weight.shape[0])
    Wg = _conv1x1_weights_for(x, weight, stride, padding, dilation, groups)
    if Wg is not None:
        return ops.conv1x1(x, Wg, weight.shape[0])
    W7 = _conv7x7_split_weights_for(x, weight, stride, padding, dilation, groups)
    if W7 is not None:
        return ops.conv7x7_split(x, W7, weight.shape[0], stride=tuple(stride)[0])
    Wg = _conv7x7_weights_for(x, weight, stride, padding, dilation, groups)
    if Wg is not None:
        return ops.conv7x7(x, Wg, weight.shape[0], stride=tuple(stride)[0])
    Us = _split_s2_weights_for(x, weight, stride, padding, dilation, groups)
    if Us is not None:
        return ops.conv3x3s2_split(x, Us, weight.shape[0])
    Wg = _conv3x3s2_weights_for(x, weight, stride, padding, dilation, groups)
    if Wg is not None:
        return ops.conv3x3s2(x, Wg, weight.shape[0])
    ks = tuple(weight.shape[2:])
    if (_hand() and ks in ((1, 5), (5, 1)) and tuple(padding) == (ks[0] // 2, ks[1] // 2) and tuple(stride) == (1, 1)
            and tuple(dilation) == (1, 1) and groups == 1 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and x.is_cuda and x.is_contiguous() and weight.shape[1] % 8 == 0 and weight.shape[0] >= 32
            and weight.shape[1] == x.shape[1]):
        key = (weight._version, weight.data_ptr())
        if split_on() and weight.shape[1] % 16 == 0 and x.shape[3] % 2 == 0:
            # the separable GRU's gate convolutions on the split direct kernel
            return ops.conv5_split(x, _split_cached(weight, "_vo_conv5_split", ops.conv5_split_filter), weight.shape[0], ks == (5, 1))
        # ... or as one-dimensional Winograd F(4,5) on the float32 matrix instructions
        cached = getattr(weight, "_vo_winograd1d", None)
        if cached is None or cached[0] != key:
            cached = (key, ops.winograd1d_filter(weight))
            weight._vo_winograd1d = cached
        return ops.conv5_winograd(x, cached[1], weight.shape[0], ks == (5, 1))
    return F.conv2d(x, weight, None, stride, padding, dilation, groups)


def conv_affine_act(x: torch.Tensor, conv: nn.Conv2d, scale, shift: torch.Tensor, relu: bool, skip=None,
                    out: torch.Tensor | None = None) -> torch.Tensor:
    """act(conv(x) * scale[c] + shift[c]) [then relu(skip + .)] with the bias-free convolution:
    inside the Winograd kernel's store when the layer is eligible, otherwise convolution +
    one in-place pass.  ``out`` (optional) is a channel slice of a wider tensor -- the
    consumer's concatenated input -- that receives the result."""
    w = weight_for(conv, x)
    U = _winograd_filter_for(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
    if U is not None and (skip is None or (skip.dtype == torch.float32 and skip.is_contiguous())):
        return ops.conv3x3_winograd(x, U, w.shape[0], scale, shift, relu, skip, out)
    if skip is None:
        Ws = _conv1x1_split_weights_for(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
        if Ws is not None:
            return ops.conv1x1_split(x, Ws, w.shape[0], scale, shift, relu, out)
        Wg = _conv1x1_weights_for(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
        if Wg is not None:
            return ops.conv1x1(x, Wg, w.shape[0], scale, shift, relu, out)
        if out is None:
            W7 = _conv7x7_split_weights_for(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
            if W7 is not None:
                return ops.conv7x7_split(x, W7, w.shape[0], scale, shift, relu, conv.stride[0])
            Wg = _conv7x7_weights_for(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
            if Wg is not None:
                return ops.conv7x7(x, Wg, w.shape[0], scale, shift, relu, conv.stride[0])
            Us = _split_s2_weights_for(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
            if Us is not None:
                return ops.conv3x3s2_split(x, Us, w.shape[0], scale, shift, relu)
            Wg = _conv3x3s2_weights_for(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
            if Wg is not None:
                return ops.conv3x3s2(x, Wg, w.shape[0], scale, shift, relu)
    y = _conv_nobias(x, conv)
    if scale is None:
        scale = getattr(conv, "_vo_ones", None)
        if scale is None or scale.device != y.device:
            scale = torch.ones(conv.out_channels, device=y.device, dtype=torch.float32)
            conv._vo_ones = scale
    y = ops.channel_affine_relu_(y, scale, shift, relu, skip)
    if out is not None:
        out.copy_(y)
        return out
    return y


def _conv_nobias(x: torch.Tensor, conv: nn.Conv2d) -> torch.Tensor:
    if conv.kernel_size == (1, 1) and conv.stride != (1, 1) and conv.padding == (0, 0):
        # a strided 1x1 convolution only ever reads the sampled pixels: gather them once and
        # run a dense GEMM instead of MIOpen's strided 1x1 path (18 TFLOP/s on these shapes)
        w = weight_for(conv, x)
        Ws = _conv1x1s2_split_weights_for(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
        if Ws is not None:
            return ops.conv1x1s2_split(x, Ws, w.shape[0])
        if (_hand() and conv.stride == (2, 2) and x.dtype == torch.float32 and w.dtype == torch.float32 and x.is_cuda
                and x.is_contiguous() and x.shape[1] % 8 == 0 and conv.groups == 1 and conv.dilation == (1, 1)
                and x.shape[1] * x.shape[2] * x.shape[3] * 4 < 2 ** 30):
            wo = (x.shape[3] - 1) // 2 + 1
            if wo >= 0.6 * ((wo + 63) // 64 * 64):               # the kernel works on 64-pixel row segments
                key = (w._version, w.data_ptr())
                cached = getattr(w, "_vo_conv1x1", None)
                if cached is None or cached[0] != key:
                    cached = (key, ops.conv1x1_filter(w))
                    w._vo_conv1x1 = cached
                return ops.conv1x1s2(x, cached[1], w.shape[0])
        x = x[:, :, ::conv.stride[0], ::conv.stride[1]].contiguous()
        return conv2d_nobias(x, w, (1, 1), (0, 0), conv.dilation, conv.groups)
    return conv2d_nobias(x, weight_for(conv, x), conv.stride, conv.padding, conv.dilation, conv.groups)


class RawNorm:
    """act(instance_norm(y)) that has not been written out: the raw convolution output ``y`` with its plane
    sums ``stats`` (N,C,2).  Consumers that can normalise while they read (the split-bf16 3x3 kernel's load,
    the residual join's skip branch) take it as it is; ``tensor()`` materialises it in place otherwise."""

    def __init__(self, y: torch.Tensor, stats: torch.Tensor, eps: float, relu: bool):
        self.y, self.stats, self.eps, self.relu = y, stats, eps, relu
        self.shape, self.dtype, self.device = y.shape, y.dtype, y.device

    def tensor(self) -> torch.Tensor:
        return ops.instance_norm_apply_(self.y, self.stats, self.eps, self.relu)


def conv_norm_act(x: torch.Tensor, conv: nn.Conv2d, norm: nn.Module, relu: bool,
                  skip: torch.Tensor | None = None, lazy: bool = False):
    """relu(norm(conv(x))) (or without the ReLU); with ``skip`` the residual join
    relu(skip + .) on top of it.  Fused path: the convolution runs WITHOUT its bias and
    one in-place HIP pass applies bias + norm + ReLU (+ join) -- instance norm removes
    any per-channel constant, so the bias drops out entirely; inference batch norm
    absorbs it into its shift.  Stock operators otherwise.  ``lazy``: when the convolution
    leaves its plane sums, return a RawNorm (the normalisation left to the consumer)."""
    if _fusable(x) and (skip is None or (skip.is_cuda and skip.dtype == x.dtype and skip.is_contiguous())):
        if isinstance(norm, nn.InstanceNorm2d) and not norm.affine and not norm.track_running_stats:
            w = weight_for(conv, x)
            y = None
            if x.dtype == torch.float32:
                W7 = _conv7x7_split_weights_for(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
                Ws = _conv1x1s2_split_weights_for(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
                if W7 is not None or Ws is not None:
                    # the stems and the projection shortcuts too leave their plane statistics
                    stats = torch.zeros((x.shape[0], w.shape[0], 2), device=x.device, dtype=torch.float64)
                    y = (ops.conv7x7_split(x, W7, w.shape[0], stride=conv.stride[0], stats=stats) if W7 is not None
                         else ops.conv1x1s2_split(x, Ws, w.shape[0], stats=stats))
                    if lazy and skip is None:
                        return RawNorm(y, stats, norm.eps, relu)
                    return ops.instance_norm_apply_(y, stats, norm.eps, relu, skip)
            U = _winograd_filter_for(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
            if U is not None:
                # the convolution's store accumulates the plane statistics: the norm is one pass
                stats = torch.zeros((x.shape[0], w.shape[0], 2), device=x.device, dtype=torch.float64)
                y = ops.conv3x3_winograd(x, U, w.shape[0], stats=stats)
                return ops.instance_norm_apply_(y, stats, norm.eps, relu, skip)
            Us = _split_s2_weights_for(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
            if Us is not None:
                # the downsampling layers too leave their plane statistics
                stats = torch.zeros((x.shape[0], w.shape[0], 2), device=x.device, dtype=torch.float64)
                y = ops.conv3x3s2_split(x, Us, w.shape[0], stats=stats)
                return ops.instance_norm_apply_(y, stats, norm.eps, relu, skip)
            return ops.instance_norm_relu_(_conv_nobias(x, conv), norm.eps, relu, skip)
        if isinstance(norm, nn.BatchNorm2d) and not norm.training and norm.track_running_stats:
            scale, shift = _folded_batch_norm(norm, conv.bias)
            return conv_affine_act(x, conv, scale, shift, relu, skip)
    y = norm(conv(x))
    y = torch.relu_(y) if relu else y
    return y if skip is None else add_act(skip, y)


def conv_act(x: torch.Tensor, conv: nn.Conv2d, out: torch.Tensor | None = None) -> torch.Tensor:
    """relu(conv(x)); fused: bias-free convolution with bias + ReLU in the kernel's store or in
    one in-place pass.  ``out``: optional channel slice that receives the result."""
    if _fusable(x) and conv.bias is not None:
        return conv_affine_act(x, conv, None, conv.bias, True, None, out)
    y = torch.relu_(conv_plain(x, conv))
    if out is not None:
        out.copy_(y)
        return out
    return y


def _folded_batch_norm(norm: nn.BatchNorm2d, conv_bias=None):
    """(scale, shift) with x*scale + shift equal to the inference-mode batch norm of
    (x + conv_bias); cached on the module and rebuilt when any tensor is modified or moved."""
    tensors = [norm.running_mean, norm.running_var] + [t for t in (norm.weight, norm.bias, conv_bias) if t is not None]
    key = tuple((t._version, t.data_ptr()) for t in tensors)
    cached = getattr(norm, "_vo_folded", None)
    if cached is None or cached[0] != key:
        w = norm.weight if norm.weight is not None else torch.ones_like(norm.running_var)
        b = norm.bias if norm.bias is not None else torch.zeros_like(norm.running_var)
        scale = (w * torch.rsqrt(norm.running_var + norm.eps)).float().contiguous()
        mean = norm.running_mean if conv_bias is None else norm.running_mean - conv_bias
        shift = (b - mean * scale).float().contiguous()
        cached = (key, scale, shift)
        norm._vo_folded = cached
    return cached[1], cached[2]


def add_act(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """relu(x + y); in place into y (a fresh conv output) when fused."""
    if _fusable(y) and x.is_cuda and x.dtype == y.dtype and x.is_contiguous() and x.shape == y.shape:
        return ops.add_relu_(y, x)
    return torch.relu(x + y)


def gray_stem(conv: nn.Conv2d, x: torch.Tensor) -> nn.Conv2d:
    """The stem convolution for a ONE-plane input.  The wrappers replicate a gray frame into three equal
    channels (flow/flow_map_RAFT.py:96,101-102); for such an input sum_c w[k,c]*x[c] = (sum_c w[k,c])*x, so
    the stem runs on the single plane with its filter summed over the input channels: a third of the
    multiplications of the networks' most expensive few-channel layers.  Returns ``conv`` itself for a
    three-channel input, else a cached single-channel twin (rebuilt when the weights change or move)."""
    if x.shape[1] == conv.in_channels or x.shape[1] != 1:
        return conv
    key = (conv.weight._version, conv.weight.data_ptr(), None if conv.bias is None else (conv.bias._version, conv.bias.data_ptr()))
    cached = getattr(conv, "_vo_gray_twin", None)
    if cached is None or cached[0] != key:
        twin = nn.Conv2d(1, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=conv.bias is not None)
        twin = twin.to(conv.weight.device)
        with torch.no_grad():
            twin.weight.copy_(conv.weight.sum(dim=1, keepdim=True))
            if conv.bias is not None:
                twin.bias.copy_(conv.bias)
        twin.eval()
        cached = (key, twin)
        object.__setattr__(conv, "_vo_gray_twin", cached)          # not a registered sub-module: it holds no state of its own
    return cached[1]


def padded_input_twin(conv: nn.Conv2d, channels: int) -> nn.Conv2d:
    """``conv`` for an input whose channel count was padded with zero planes to ``channels`` (a multiple of 16,
    so that the layer can run on the split-bf16 kernels): a cached twin whose weights are zero-extended along
    the input channels (rebuilt when the weights change or move).  Returns ``conv`` itself if nothing is padded."""
    if channels == conv.in_channels:
        return conv
    if channels < conv.in_channels:
        raise ValueError("the padded channel count is smaller than the layer's")
    key = (channels, conv.weight._version, conv.weight.data_ptr(), None if conv.bias is None else (conv.bias._version, conv.bias.data_ptr()))
    cached = getattr(conv, "_vo_padded_twin", None)
    if cached is None or cached[0] != key:
        twin = nn.Conv2d(channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=conv.bias is not None)
        twin = twin.to(conv.weight.device)
        with torch.no_grad():
            twin.weight.zero_()
            twin.weight[:, :conv.in_channels].copy_(conv.weight)
            if conv.bias is not None:
                twin.bias.copy_(conv.bias)
        twin.eval()
        cached = (key, twin)
        object.__setattr__(conv, "_vo_padded_twin", cached)
    return cached[1]


def _norm(kind: str, planes: int) -> nn.Module:
    if kind == "group":
        return nn.GroupNorm(num_groups=planes // 8, num_channels=planes)
    if kind == "batch":
        return nn.BatchNorm2d(planes)
    if kind == "instance":
        return nn.InstanceNorm2d(planes)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(f"unknown norm '{kind}'")


def _shortcut_fuses(x: torch.Tensor, conv3x3: nn.Conv2d, conv1x1: nn.Conv2d) -> bool:
    """True when a stride-2 block's 3x3 convolution and its projection shortcut can share one pass over x
    (ops.conv3x3s2_shortcut_split): same widths, an output width in multiples of 4, the split-bf16 arithmetic."""
    Wo = (x.shape[3] + 1) // 2
    return (split_on() and x.is_contiguous() and conv3x3.out_channels == conv1x1.out_channels and x.shape[3] % 2 == 0
            and Wo % 4 == 0 and conv3x3.stride == (2, 2) and conv1x1.stride == (2, 2))


class ResidualBlock(nn.Module):
    """Two 3x3 convs with a (strided 1x1) skip; the skip conv exists whenever the
    block changes resolution or width."""

    def __init__(self, in_planes: int, planes: int, norm_fn: str = "group", stride: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = _norm(norm_fn, planes)
        self.norm2 = _norm(norm_fn, planes)
        if stride == 1 and in_planes == planes:
            self.downsample = None
        else:
            self.norm3 = _norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def forward(self, x):
        """x: a tensor, or a RawNorm (the stem's output still waiting for its normalisation)."""
        chained = self._instance_norm_chain(x)
        if chained is not None:
            return chained
        if isinstance(x, RawNorm):
            x = x.tensor()
        fused = self._batch_norm_downsampling_pair(x)
        if fused is not None:
            y, x = fused
        else:
            y = conv_norm_act(x, self.conv1, self.norm1, True)
            if self.downsample is not None:
                x = conv_norm_act(x, self.downsample[0], self.norm3, False)
        return conv_norm_act(y, self.conv2, self.norm2, True, skip=x)        # relu(x + relu(norm2(conv2(y))))

    def _batch_norm_downsampling_pair(self, x):
        """(relu(norm1(conv1(x))), norm3(shortcut(x))) of a stride-2 block with inference batch norms from ONE
        pass over x (the norms folded into the two stores), or None where that kernel does not apply."""
        if self.downsample is None or not (_fusable(x) and x.dtype == torch.float32):
            return None
        bn = lambda n: isinstance(n, nn.BatchNorm2d) and not n.training and n.track_running_stats
        c1, d = self.conv1, self.downsample[0]
        if not (bn(self.norm1) and bn(self.norm3) and _shortcut_fuses(x, c1, d)):
            return None
        Us = _split_s2_weights_for(x, c1.weight, c1.stride, c1.padding, c1.dilation, c1.groups)
        Ws = _conv1x1s2_split_weights_for(x, d.weight, d.stride, d.padding, d.dilation, d.groups)
        if Us is None or Ws is None:
            return None
        s1, b1 = _folded_batch_norm(self.norm1, c1.bias)
        s3, b3 = _folded_batch_norm(self.norm3, d.bias)
        return ops.conv3x3s2_shortcut_split(x, Us, Ws, c1.out_channels, s1, b1, True, None, s3, b3, False, None)

    def _instance_norm_chain(self, x):
        """Both 3x3 convolutions on the matrix-core kernels with instance norm: conv1's store leaves
        the plane sums of its raw output, conv2 normalises + ReLUs that output while loading
        it and leaves its own sums, and one pass applies norm2 + ReLU + the residual join.
        The normalised tensor between the two convolutions is never written.  The same holds for the
        block's input when it arrives as a RawNorm (conv1 and the join normalise it as they read) and for a
        projection shortcut (its 1x1 stride-2 convolution leaves its sums, the join normalises it)."""
        plain = lambda n: isinstance(n, nn.InstanceNorm2d) and not n.affine and not n.track_running_stats
        raw = x if isinstance(x, RawNorm) else None
        xt = raw.y if raw is not None else x
        if not (_fusable(xt) and xt.dtype == torch.float32 and plain(self.norm1) and plain(self.norm2)):
            return None
        c1, c2 = self.conv1, self.conv2
        if c2.in_channels > 256 or (raw is not None and (not raw.relu or c1.in_channels > 256)):
            return None
        st1 = torch.zeros((xt.shape[0], c1.out_channels, 2), device=xt.device, dtype=torch.float64)
        if self.downsample is None:
            U1 = _winograd_filter_for(xt, c1.weight, c1.stride, c1.padding, c1.dilation, c1.groups)
            if U1 is None:
                return None
            if raw is not None:
                y1 = ops.conv3x3_winograd(xt, U1, c1.out_channels, stats=st1, in_stats=raw.stats, in_eps=raw.eps)
            else:
                y1 = ops.conv3x3_winograd(xt, U1, c1.out_channels, stats=st1)
            skip, skip_stats = xt, (raw.stats if raw is not None else None)
            skip_eps, skip_relu = (raw.eps if raw is not None else 1e-5), True
        else:
            # stride-2 block: conv1 on the stride-2 kernel, the shortcut's 1x1 stride-2 convolution with its own sums
            if raw is not None or not plain(self.norm3):
                return None
            d = self.downsample[0]
            Us = _split_s2_weights_for(xt, c1.weight, c1.stride, c1.padding, c1.dilation, c1.groups)
            Ws = _conv1x1s2_split_weights_for(xt, d.weight, d.stride, d.padding, d.dilation, d.groups)
            if Us is None or Ws is None:
                return None
            skip_stats = torch.zeros((xt.shape[0], d.out_channels, 2), device=xt.device, dtype=torch.float64)
            if _shortcut_fuses(xt, c1, d):
                # one pass over the block's input: the shortcut's samples are the 3x3 filter's centre tap
                y1, skip = ops.conv3x3s2_shortcut_split(xt, Us, Ws, c1.out_channels, stats=st1, stats2=skip_stats)
            else:
                y1 = ops.conv3x3s2_split(xt, Us, c1.out_channels, stats=st1)
                skip = ops.conv1x1s2_split(xt, Ws, d.out_channels, stats=skip_stats)
            skip_eps, skip_relu = self.norm3.eps, False
        U2 = _winograd_filter_for(y1, c2.weight, c2.stride, c2.padding, c2.dilation, c2.groups)
        if U2 is None:
            y1 = ops.instance_norm_apply_(y1, st1, self.norm1.eps, True)
            if skip_stats is not None:
                skip = ops.instance_norm_apply_(skip, skip_stats, skip_eps, skip_relu)
            return conv_norm_act(y1, c2, self.norm2, True, skip=skip)
        st2 = torch.zeros((xt.shape[0], c2.out_channels, 2), device=xt.device, dtype=torch.float64)
        y2 = ops.conv3x3_winograd(y1, U2, c2.out_channels, stats=st2, in_stats=st1, in_eps=self.norm1.eps)
        return ops.instance_norm_apply_(y2, st2, self.norm2.eps, True, skip, skip_stats, skip_eps, skip_relu)


def _init_weights(module: nn.Module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
            if m.weight is not None:
                nn.init.constant_(m.weight, 1)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


class BasicEncoder(nn.Module):
    """Feature / context encoder: 7x7 stem then three residual stages (64, 96, 128)
    and a 1x1 projection.  ``downsample`` = 3 gives 1/8 resolution (RAFT); 2 gives
    1/4 (RAFT-Stereo with n_downsample=2: the stem runs at full resolution)."""

    def __init__(self, output_dim: int = 128, norm_fn: str = "batch", downsample: int = 3):
        super().__init__()
        self.norm_fn = norm_fn
        self.norm1 = _norm(norm_fn, 64)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=1 + (downsample > 2), padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = 64
        self.layer1 = self._make_layer(64, stride=1)
        self.layer2 = self._make_layer(96, stride=1 + (downsample > 1))
        self.layer3 = self._make_layer(128, stride=1 + (downsample > 0))
        self.conv2 = nn.Conv2d(128, output_dim, kernel_size=1)
        _init_weights(self)

    def _make_layer(self, dim: int, stride: int = 1) -> nn.Sequential:
        blocks = nn.Sequential(ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride),
                               ResidualBlock(dim, dim, self.norm_fn, stride=1))
        self.in_planes = dim
        return blocks

    def forward(self, x: torch.Tensor, pixel_major: bool = False, split_rows: bool = False) -> torch.Tensor:
        """``pixel_major``: return float32 (B,H,W,D) instead of (B,D,H,W); with ``split_rows`` the
        (B,H,W,D) rows as bf16 pieces (see conv_pixel_major).  x: three channels, or the one plane of a
        gray frame."""
        x = conv_norm_act(x, gray_stem(self.conv1, x), self.norm1, True, lazy=True)     # a RawNorm when the stem leaves its sums
        x = self.layer3(self.layer2(self.layer1(x)))
        return conv_pixel_major(x, self.conv2, split_rows) if pixel_major else conv_plain(x, self.conv2)


class MultiBasicEncoder(nn.Module):
    """RAFT-Stereo context encoder: the same trunk plus two more stride-2 stages,
    with one (hidden, context) head pair per GRU resolution."""

    def __init__(self, output_dim, norm_fn: str = "batch", downsample: int = 3):
        super().__init__()
        self.norm_fn = norm_fn
        self.norm1 = _norm(norm_fn, 64)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=1 + (downsample > 2), padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = 64
        self.layer1 = self._make_layer(64, stride=1)
        self.layer2 = self._make_layer(96, stride=1 + (downsample > 1))
        self.layer3 = self._make_layer(128, stride=1 + (downsample > 0))
        self.layer4 = self._make_layer(128, stride=2)
        self.layer5 = self._make_layer(128, stride=2)
        self.outputs08 = nn.ModuleList([
            nn.Sequential(ResidualBlock(128, 128, norm_fn, stride=1), nn.Conv2d(128, dim[2], 3, padding=1))
            for dim in output_dim])
        self.outputs16 = nn.ModuleList([
            nn.Sequential(ResidualBlock(128, 128, norm_fn, stride=1), nn.Conv2d(128, dim[1], 3, padding=1))
            for dim in output_dim])
        self.outputs32 = nn.ModuleList([nn.Conv2d(128, dim[0], 3, padding=1) for dim in output_dim])
        _init_weights(self)

    def _make_layer(self, dim: int, stride: int = 1) -> nn.Sequential:
        blocks = nn.Sequential(ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride),
                               ResidualBlock(dim, dim, self.norm_fn, stride=1))
        self.in_planes = dim
        return blocks

    def forward(self, x: torch.Tensor, num_layers: int = 3):
        x = conv_norm_act(x, gray_stem(self.conv1, x), self.norm1, True)
        x = self.layer3(self.layer2(self.layer1(x)))
        out08 = [conv_plain(f[0](x), f[1]) for f in self.outputs08]
        if num_layers == 1:
            return (out08,)
        y = self.layer4(x)
        out16 = [conv_plain(f[0](y), f[1]) for f in self.outputs16]
        if num_layers == 2:
            return (out08, out16)
        z = self.layer5(y)
        out32 = [conv_plain(z, f) for f in self.outputs32]
        return (out08, out16, out32)
