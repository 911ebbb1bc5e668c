"""RAFT-Flow and RAFT-Stereo on hand-written HIP kernels.

The all-pairs correlation volume, its pyramid, the per-iteration windowed lookup and the final
convex upsampling go through the C ABI in ``visual_odometry_amd.ops``; the conv encoders and GRUs
are ``nn.Module`` trees that hold the parameters under the published names and dispatch their
convolutions, norms and gate arithmetic to hand-written kernels of the same library (raft/layers.py,
raft/update.py).  Nothing here runs without a GPU.  Both models are inference-only (``test_mode`` of
the published models): they return the low-resolution field and the upsampled field of the LAST
iteration, and evaluate the mask head on that iteration only.  ``conv_arithmetic`` (a field of each
model: "f16x3", "bf16x3" or "f32", see layers.Arithmetic) says how the heavy products are formed.

Argument sets: RAFT-Flow as flow/flow_map_RAFT.py:36-40 (full model, 4 levels,
radius 4); RAFT-Stereo as stereo/stereo_disparity_RAFT.py:33-44, with
``hidden_dims`` / ``n_gru_layers`` overridable for the "realtime" variant of
stereo/raft_args_realtime.json:9,17.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn as nn

from .. import ops
from .layers import CONV_ARITHMETICS, BasicEncoder, MultiBasicEncoder, arithmetic, conv_plain
from .update import BasicMultiUpdateBlock, BasicUpdateBlock


def coords_grid(batch: int, ht: int, wd: int, device) -> torch.Tensor:
    ys, xs = torch.meshgrid(torch.arange(ht, device=device, dtype=torch.float32),
                            torch.arange(wd, device=device, dtype=torch.float32), indexing="ij")
    return torch.stack([xs, ys], dim=0)[None].repeat(batch, 1, 1, 1).contiguous()


def normalize_image(img: torch.Tensor) -> torch.Tensor:
    """0..255 -> -1..1, as both published models do on entry."""
    return (2 * (img / 255.0) - 1.0).contiguous()


def pixel_major(fmap: torch.Tensor) -> torch.Tensor:
    """(B,D,H,W) -> (B,H,W,D) contiguous float32: the layout the correlation kernel
    reads (each pixel's feature vector is one contiguous K-row)."""
    return fmap.float().permute(0, 2, 3, 1).contiguous()


def _checked(kind: str) -> str:
    if kind not in CONV_ARITHMETICS:
        raise ValueError(f"conv_arithmetic must be one of {CONV_ARITHMETICS}, not {kind!r}")
    return kind


def load_checkpoint(model: nn.Module, path: str, device) -> None:
    """Load a state dict saved from ``nn.DataParallel(model)`` (keys prefixed with
    ``module.``), which is how the published checkpoints are stored and why the
    reference wraps its models before loading (flow/flow_map_RAFT.py:42-45)."""
    state = torch.load(path, map_location=device, weights_only=True)
    state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}
    model.load_state_dict(state)


# --------------------------------------------------------------------------- flow

def _flow_volume_f16(fmap1: torch.Tensor, fmap2: torch.Tensor) -> torch.Tensor:
    """The flow volume on float16 pieces: every image's map is scaled by its own power of two, cut once
    (ops.split_feature_rows) and multiplied by the kernel that reads cut rows.  When fmap2 is fmap1 shifted by one
    frame inside one buffer (consecutive pairs of a sequence: maps[:-1], maps[1:]) the B+1 distinct maps are cut
    once instead of 2B."""
    B = fmap1.shape[0]
    fmap1, fmap2 = fmap1.contiguous(), fmap2.contiguous()
    per_frame = fmap1[0].numel()
    if (fmap2.data_ptr() == fmap1.data_ptr() + per_frame * 4 and fmap1.untyped_storage().data_ptr() == fmap2.untyped_storage().data_ptr()
            and fmap1.storage_offset() + (B + 1) * per_frame <= fmap1.untyped_storage().nbytes() // 4):
        maps = torch.as_strided(fmap1, (B + 1,) + tuple(fmap1.shape[1:]), fmap1.stride())
        amax = ops.feature_rows_amax(maps)
        cut = ops.split_feature_rows(maps, 32, "f16", amax)
        return ops.corr_volume_2d_presplit(cut[:B], cut[1:], "f16", amax[:B], amax[1:])
    a1, a2 = ops.feature_rows_amax(fmap1), ops.feature_rows_amax(fmap2)
    return ops.corr_volume_2d_presplit(ops.split_feature_rows(fmap1, 32, "f16", a1), ops.split_feature_rows(fmap2, 32, "f16", a2),
                                       "f16", a1, a2)


@dataclass
class RAFTFlowConfig:
    corr_levels: int = 4
    corr_radius: int = 4
    hidden_dim: int = 128
    context_dim: int = 128


class RAFTFlow(nn.Module):
    """RAFT optical flow at 1/8 resolution, 8x convex upsampling."""

    def __init__(self, cfg: Optional[RAFTFlowConfig] = None):
        super().__init__()
        self.cfg = cfg or RAFTFlowConfig()
        c = self.cfg
        self.fnet = BasicEncoder(output_dim=256, norm_fn="instance", downsample=3)
        self.cnet = BasicEncoder(output_dim=c.hidden_dim + c.context_dim, norm_fn="batch", downsample=3)
        self.update_block = BasicUpdateBlock(c.corr_levels, c.corr_radius, hidden_dim=c.hidden_dim)
        # The published models' ``mixed_precision`` switch (off in the reference,
        # flow/flow_map_RAFT.py:38): float16 activations in the conv nets over float32
        # master weights; correlation, lookup, coordinates and upsampling stay float32.
        self.mixed_precision = False
        self.conv_arithmetic = "f16x3"                       # see layers.Arithmetic.kind

    def _act_dtype(self):
        return torch.float16 if self.mixed_precision else torch.float32

    def _scope(self):
        """The arithmetic of this model's forward passes: its own ``conv_arithmetic``, fused kernels on."""
        return arithmetic(_checked(self.conv_arithmetic), fused=True)

    def presplit_features(self) -> bool:
        """True when features() hands the correlation kernel rows that are already split into bf16
        pieces (ops.split_feature_rows(., 32)): under the "bf16x3" arithmetic, outside mixed precision.
        Such maps are bytes for that kernel, not float32 numbers."""
        return self.conv_arithmetic == "bf16x3" and not self.mixed_precision

    @torch.no_grad()
    def features(self, images: torch.Tensor) -> torch.Tensor:
        """Normalised images (B,3,H,W) -> pixel-major feature maps (B,H/8,W/8,256); see presplit_features."""
        with self._scope():
            return self.fnet(images.to(self._act_dtype()), pixel_major=True, split_rows=self.presplit_features())

    @torch.no_grad()
    def forward(self, image1, image2, iters: int = 12, flow_init=None, fmap1=None, fmap2=None,
                normalized: bool = False):
        """image1/2: (B,3,H,W) float 0..255 (or already in [-1,1] with
        ``normalized=True``), H and W multiples of 8.  Precomputed feature maps
        (what features() returns under the current arithmetic) may be passed so that
        consecutive pairs of a sequence share the encoder pass of their common frame."""
        c = self.cfg
        if not normalized:
            image1 = normalize_image(image1)
        if fmap1 is None or fmap2 is None:
            if not normalized:
                image2 = normalize_image(image2)
            both = self.features(torch.cat([image1, image2], dim=0))
            fmap1, fmap2 = both[: image1.shape[0]], both[image1.shape[0]:]
        B, H8, W8, _ = fmap1.shape

        # row-major target order: a tiled order (8x4-pixel tiles per 128-byte line) gives identical values but
        # was measured slower and heavier on the memory counters for the lookup's gathers (tools/tiled_volume.hip.txt)
        if self.presplit_features():
            corr = ops.corr_volume_2d_presplit(fmap1.contiguous(), fmap2.contiguous())
        elif self.conv_arithmetic == "f16x3" and fmap1.dtype == torch.float32 and fmap1.shape[-1] % 64 == 0:
            corr = _flow_volume_f16(fmap1, fmap2)
        else:
            corr = ops.corr_volume_2d(fmap1, fmap2, split=self.conv_arithmetic == "bf16x3")
        pyramid = ops.corr_pyramid_2d(corr, c.corr_levels)

        act = self._act_dtype()
        with self._scope():
            net, inp = torch.split(self.cnet(image1.to(act)), [c.hidden_dim, c.context_dim], dim=1)
            net, inp = torch.tanh(net), torch.relu(inp)

        coords0 = coords_grid(B, H8, W8, image1.device)
        coords1 = coords0.clone()
        if flow_init is not None:
            coords1 = coords1 + flow_init
        raw_mask = None
        state = {}                                               # the GRU's input buffer lives across iterations
        # the window's 324 channels are padded with zero planes to a multiple of 16 when the consuming 1x1
        # convolution runs on the split-bf16 kernel (which takes its input channels in chunks of 16)
        win_ch = c.corr_levels * (2 * c.corr_radius + 1) ** 2
        pad_ch = (-win_ch) % 16 if (self.conv_arithmetic != "f32" and act == torch.float32 and W8 % 2 == 0) else 0
        window_buf = torch.empty((B, win_ch + pad_ch, H8, W8), device=image1.device, dtype=torch.float32)
        if pad_ch:
            window_buf[:, win_ch:].zero_()
        for it in range(iters):
            ops.corr_lookup_2d(pyramid, coords1.contiguous(), c.corr_radius, out=window_buf[:, :win_ch])
            window = window_buf
            flow = coords1 - coords0
            with self._scope():
                net, raw_mask, delta = self.update_block(net, inp, window.to(act), flow.to(act),
                                                         want_mask=(it == iters - 1), state=state)
            coords1 = coords1 + delta.float()
        flow_low = (coords1 - coords0).contiguous()
        flow_up = ops.convex_upsample(flow_low, raw_mask.float().contiguous(), 8, mask_scale=0.25)
        return flow_low, flow_up


# ------------------------------------------------------------------------- stereo

@dataclass
class RAFTStereoConfig:
    hidden_dims: List[int] = field(default_factory=lambda: [128, 128, 128])
    n_gru_layers: int = 3
    n_downsample: int = 2
    corr_levels: int = 4
    corr_radius: int = 4
    context_norm: str = "batch"
    slow_fast_gru: bool = False

    @staticmethod
    def realtime() -> "RAFTStereoConfig":
        return RAFTStereoConfig(hidden_dims=[64, 64, 64], n_gru_layers=2)


class RAFTStereo(nn.Module):
    """RAFT-Stereo with the "reg" (materialised per-row volume) correlation."""

    def __init__(self, cfg: Optional[RAFTStereoConfig] = None):
        super().__init__()
        self.cfg = cfg or RAFTStereoConfig()
        c = self.cfg
        context_dims = c.hidden_dims
        self.cnet = MultiBasicEncoder(output_dim=[c.hidden_dims, context_dims], norm_fn=c.context_norm,
                                      downsample=c.n_downsample)
        self.update_block = BasicMultiUpdateBlock(c.hidden_dims, c.n_gru_layers, c.n_downsample,
                                                  c.corr_levels, c.corr_radius)
        self.context_zqr_convs = nn.ModuleList(
            [nn.Conv2d(context_dims[i], c.hidden_dims[i] * 3, 3, padding=1) for i in range(c.n_gru_layers)])
        self.fnet = BasicEncoder(output_dim=256, norm_fn="instance", downsample=c.n_downsample)
        self.mixed_precision = False     # as in RAFTFlow (stereo/stereo_disparity_RAFT.py:43 leaves it off)
        self.conv_arithmetic = "f16x3"   # see layers.Arithmetic.kind

    def _scope(self):
        return arithmetic(_checked(self.conv_arithmetic), fused=True)

    @torch.no_grad()
    def forward(self, image1, image2, iters: int = 12, flow_init=None, normalized: bool = False):
        """image1/2: (B,3,H,W) float 0..255 (or already in [-1,1] with
        ``normalized=True``), H and W multiples of 32.  Returns (low-res x-flow field
        (B,2,h,w), upsampled disparity (B,1,H,W), negative for a rectified
        left/right pair)."""
        c = self.cfg
        if not normalized:
            image1 = normalize_image(image1)
            image2 = normalize_image(image2)
        B = image1.shape[0]

        act = torch.float16 if self.mixed_precision else torch.float32
        with self._scope():
            image1, image2 = image1.to(act), image2.to(act)
            cnet_list = self.cnet(image1, num_layers=c.n_gru_layers)
            both = self.fnet(torch.cat([image1, image2], dim=0), pixel_major=True)
            net_list = [torch.tanh(x[0]) for x in cnet_list]
            inp_list = [torch.relu(x[1]) for x in cnet_list]
            # the GRUs' context terms do not change across iterations: convolve once
            inp_list = [list(conv_plain(i, conv).split(split_size=conv.out_channels // 3, dim=1))
                        for i, conv in zip(inp_list, self.context_zqr_convs)]
        fmap1, fmap2 = both[:B], both[B:]

        if self.conv_arithmetic == "f16x3" and both.dtype == torch.float32:
            amax = ops.feature_rows_amax(both)                   # per image, stays on the device
            corr = ops.corr_volume_1d(fmap1, fmap2, split="f16", amax1=amax[:B], amax2=amax[B:])
        else:
            corr = ops.corr_volume_1d(fmap1, fmap2, split=self.conv_arithmetic == "bf16x3")
        pyramid = ops.corr_pyramid_1d(corr, c.corr_levels)

        _, _, h, w = net_list[0].shape
        coords0 = coords_grid(B, h, w, image1.device)
        coords1 = coords0.clone()
        if flow_init is not None:
            coords1 = coords1 + flow_init
        raw_mask = None
        state = {}                                               # GRU input buffers live across iterations
        for it in range(iters):
            window = ops.corr_lookup_1d(pyramid, coords1.contiguous(), c.corr_radius)
            flow = coords1 - coords0
            with self._scope():
                if c.n_gru_layers == 3 and c.slow_fast_gru:
                    net_list = self.update_block(net_list, inp_list, iter32=True, iter16=False, iter08=False,
                                                 update=False, state=state)
                if c.n_gru_layers >= 2 and c.slow_fast_gru:
                    net_list = self.update_block(net_list, inp_list, iter32=c.n_gru_layers == 3, iter16=True,
                                                 iter08=False, update=False, state=state)
                net_list, raw_mask, delta = self.update_block(net_list, inp_list, window.to(act), flow.to(act),
                                                              iter32=c.n_gru_layers == 3, iter16=c.n_gru_layers >= 2,
                                                              want_mask=(it == iters - 1), state=state)
            delta = delta.float()
            delta = torch.cat([delta[:, :1], torch.zeros_like(delta[:, 1:])], dim=1)   # epipolar: no y motion
            coords1 = coords1 + delta
        flow_low = (coords1 - coords0).contiguous()
        disparity = ops.convex_upsample(flow_low, raw_mask.float().contiguous(), 2 ** c.n_downsample, out_channels=1,
                                        mask_scale=0.25)
        return flow_low, disparity
