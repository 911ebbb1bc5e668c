"""Stock-PyTorch float32 restatement of the RAFT / RAFT-Stereo hot-path operators.

TEST INFRASTRUCTURE ONLY (checker for the HIP kernels, and the thing timed as the
CPU baseline); the product package never imports it.

PARITY UNPINNED.  The reference runs these operators from two git submodules,
``external/RAFT-Flow`` (package ``flow_core``, a fork of princeton-vl/RAFT) and
``external/RAFT-Stereo`` (package ``stereo_core``, a fork of
princeton-vl/RAFT-Stereo); both directories are EMPTY in the reference checkout,
no commit is pinned (.gitmodules:1-9 has no SHA) and the reference holds no test,
golden vector or checkpoint for them.  What follows restates the published
algorithm with the argument sets the reference's call sites fix:
flow/flow_map_RAFT.py:36-40,81 (small=False, alternate_corr=False -> 4 levels,
radius 4, all-pairs volume) and stereo/stereo_disparity_RAFT.py:33-44,91
(corr_implementation="reg", corr_levels=4, corr_radius=4, n_downsample=2).
It is written with matmul / avg_pool2d / grid_sample / unfold+softmax, i.e. the
very operators the upstream models execute, so the HIP kernels are checked
operator by operator against what the reference would compute.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ sampling

def bilinear_sampler(img: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """grid_sample in pixel coordinates: coords[..., 0] = x, coords[..., 1] = y,
    normalised with (size-1) and sampled with align_corners=True, zero padding."""
    H, W = img.shape[-2:]
    xg, yg = coords.split([1, 1], dim=-1)
    xg = 2 * xg / (W - 1) - 1
    if H > 1:
        yg = 2 * yg / (H - 1) - 1
    return F.grid_sample(img, torch.cat([xg, yg], dim=-1), align_corners=True)


def coords_grid(batch: int, ht: int, wd: int, device=None) -> torch.Tensor:
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


# ---------------------------------------------------------- 2-D correlation

def corr_volume_2d(fmap1: torch.Tensor, fmap2: torch.Tensor) -> torch.Tensor:
    """(B,D,H,W) x2 -> (B, H*W, H, W): <f1[:, :, p1], f2[:, :, p2]> / sqrt(D)."""
    B, D, H, W = fmap1.shape
    a = fmap1.reshape(B, D, H * W)
    b = fmap2.reshape(B, D, H * W)
    corr = torch.matmul(a.transpose(1, 2), b)
    return corr.reshape(B, H * W, H, W) / torch.sqrt(torch.tensor(D).float())


def corr_pyramid_2d(corr: torch.Tensor, levels: int = 4):
    B, N, H, W = corr.shape
    c = corr.reshape(B * N, 1, H, W)
    pyr = [c]
    for _ in range(levels - 1):
        c = F.avg_pool2d(c, 2, stride=2)
        pyr.append(c)
    return [p.reshape(B, N, p.shape[-2], p.shape[-1]) for p in pyr]


def corr_lookup_2d(pyr, coords: torch.Tensor, radius: int = 4) -> torch.Tensor:
    """coords (B,2,H,W) -> (B, L*(2r+1)^2, H, W).  The window offsets are
    stack(meshgrid(d, d)) ADDED to (x, y): the first window index steps x."""
    r = radius
    B, _, H, W = coords.shape
    c = coords.permute(0, 2, 3, 1)
    d = torch.linspace(-r, r, 2 * r + 1, device=coords.device)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
    out = []
    for i, lvl in enumerate(pyr):
        img = lvl.reshape(B * H * W, 1, lvl.shape[-2], lvl.shape[-1])
        centre = c.reshape(B * H * W, 1, 1, 2) / 2 ** i
        s = bilinear_sampler(img, centre + delta)
        out.append(s.view(B, H, W, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


# ---------------------------------------------------------- 1-D correlation

def corr_volume_1d(fmap1: torch.Tensor, fmap2: torch.Tensor) -> torch.Tensor:
    """(B,D,H,W1),(B,D,H,W2) -> (B,H,W1,W2) per-row correlation / sqrt(D)."""
    D = fmap1.shape[1]
    corr = torch.einsum("aijk,aijh->ajkh", fmap1, fmap2)
    return corr.contiguous() / torch.sqrt(torch.tensor(D).float())


def corr_pyramid_1d(corr: torch.Tensor, levels: int = 4):
    B, H, W1, W2 = corr.shape
    c = corr.reshape(B * H * W1, 1, 1, W2)
    pyr = [c]
    for _ in range(levels - 1):
        c = F.avg_pool2d(c, [1, 2], stride=[1, 2])
        pyr.append(c)
    return [p.reshape(B, H, W1, p.shape[-1]) for p in pyr]


def corr_lookup_1d(pyr, coords: torch.Tensor, radius: int = 4) -> torch.Tensor:
    """coords (B,2,H,W), only x is used -> (B, L*(2r+1), H, W)."""
    r = radius
    B, _, H, W = coords.shape
    x = coords[:, :1].permute(0, 2, 3, 1)
    dx = torch.linspace(-r, r, 2 * r + 1, device=coords.device).view(2 * r + 1, 1)
    out = []
    for i, lvl in enumerate(pyr):
        img = lvl.reshape(B * H * W, 1, 1, lvl.shape[-1])
        x0 = dx + x.reshape(B * H * W, 1, 1, 1) / 2 ** i
        s = bilinear_sampler(img, torch.cat([x0, torch.zeros_like(x0)], dim=-1))
        out.append(s.view(B, H, W, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


# ----------------------------------------------------------- convex upsample

def convex_upsample(flow: torch.Tensor, mask: torch.Tensor, factor: int) -> torch.Tensor:
    """flow (B,C,H,W), mask (B, 9*f*f, H, W) -> (B, C, f*H, f*W)."""
    B, C, H, W = flow.shape
    m = torch.softmax(mask.view(B, 1, 9, factor, factor, H, W), dim=2)
    up = F.unfold(factor * flow, [3, 3], padding=1).view(B, C, 9, 1, 1, H, W)
    up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(B, C, factor * H, factor * W)


# ------------------------------------------------------------- input padding

def pad_to_multiple(x: torch.Tensor, divis_by: int = 8):
    """InputPadder('sintel' mode): replicate-pad H and W up to a multiple of
    ``divis_by``, split evenly (extra pixel at the end).  Returns (padded, pads)."""
    ht, wd = x.shape[-2:]
    ph = (((ht // divis_by) + 1) * divis_by - ht) % divis_by
    pw = (((wd // divis_by) + 1) * divis_by - wd) % divis_by
    pads = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
    return F.pad(x, pads, mode="replicate"), pads


def unpad(x: torch.Tensor, pads):
    ht, wd = x.shape[-2:]
    return x[..., pads[2]:ht - pads[3], pads[0]:wd - pads[1]]


# ------------------------------------------------------------ whole networks
# The module trees (encoders, motion encoders, GRUs, heads) are restated with stock PyTorch
# operators in oracle/raft_modules.py; nothing of the product's network code runs here.  A model
# handed to these functions only lends its configuration and its ``state_dict()``, which is loaded
# (strict) into the restated tree on the device and in the dtype float32 the restatement runs in.

def _twin(model, make, device):
    from . import raft_modules
    if isinstance(model, (raft_modules.RAFT, raft_modules.RAFTStereo)):
        return model
    key = tuple((k, v._version, v.data_ptr()) for k, v in model.state_dict().items())
    cached = getattr(model, "_oracle_twin", None)
    if cached is None or cached[0] != (key, str(device)):
        cached = ((key, str(device)), make(model, device))
        object.__setattr__(model, "_oracle_twin", cached)
    return cached[1]


@torch.no_grad()
def raft_flow_forward(model, image1, image2, iters=12, flow_init=None):
    """(flow at 1/8 resolution, full-resolution flow) from the restated RAFT carrying ``model``'s weights."""
    from . import raft_modules
    return _twin(model, raft_modules.flow_twin, image1.device)(image1.float(), image2.float(), iters, flow_init)


@torch.no_grad()
def raft_stereo_forward(model, image1, image2, iters=12, flow_init=None):
    """(x-flow field at working resolution, full-resolution disparity) from the restated RAFT-Stereo."""
    from . import raft_modules
    return _twin(model, raft_modules.stereo_twin, image1.device)(image1.float(), image2.float(), iters, flow_init)
