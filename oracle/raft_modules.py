"""Stock-PyTorch restatement of the RAFT and RAFT-Stereo module trees (encoders, motion
encoders, GRUs, heads) and of their inference forward passes.

TEST INFRASTRUCTURE ONLY: the checker for the product's networks and the thing timed as the
CPU baseline.  Nothing here imports the product package; a product model is compared by
loading its ``state_dict()`` into the matching class below (``strict=True``: the parameter
names must agree) and running both on the same frames.

PARITY UNPINNED.  The reference constructs these networks from two git submodules,
``flow_core.raft.RAFT(args)`` (flow/flow_map_RAFT.py:13-15,36-42; a fork of princeton-vl/RAFT)
and ``stereo_core.raft_stereo.RAFTStereo(args)`` (stereo/stereo_disparity_RAFT.py:14-15,33-46;
a fork of princeton-vl/RAFT-Stereo); both directories are empty in the reference checkout and
no commit is pinned, so the trees below restate the PUBLISHED architectures with the argument
sets those call sites fix:
  RAFT        small=False, mixed_precision=False, alternate_corr=False  (4 levels, radius 4,
              hidden 128, context 128, 1/8 resolution, 8x convex upsampling)
  RAFT-Stereo hidden_dims, n_gru_layers, n_downsample=2, context_norm="batch",
              corr_implementation="reg", corr_levels=4, corr_radius=4, shared_backbone=False,
              slow_fast_gru=False, mixed_precision=False
Sub-module and parameter names are the published ones (``fnet.layer1.0.conv1``,
``update_block.gru.convz1``, ``cnet.outputs08.0.1``, ``context_zqr_convs.0`` ...), which is what
lets a ``module.``-prefixed checkpoint of the published models load
(flow/flow_map_RAFT.py:42-47, stereo/stereo_disparity_RAFT.py:46-51).
Every operator is the stock one (nn.Conv2d, nn.InstanceNorm2d / BatchNorm2d, F.avg_pool2d,
F.interpolate, torch.sigmoid / tanh) in float32.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import raft_ref as R


# ------------------------------------------------------------------ encoders

def _norm(kind: str, planes: int) -> nn.Module:
    if kind == "group":
        return nn.GroupNorm(num_groups=planes // 8, num_channels=planes)
    if kind == "batch":
        return nn.BatchNorm2d(planes)
    if kind == "instance":
        return nn.InstanceNorm2d(planes)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(kind)


class ResidualBlock(nn.Module):
    """relu(x' + relu(norm2(conv2(relu(norm1(conv1(x))))))) with x' = x, or norm3(1x1 strided conv(x))
    when the block changes width or resolution."""

    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = _norm(norm_fn, planes)
        self.norm2 = _norm(norm_fn, planes)
        self.downsample = None
        if not (stride == 1 and in_planes == planes):
            self.norm3 = _norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def forward(self, x):
        y = self.relu(self.norm1(self.conv1(x)))
        y = self.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


def _stage(in_planes, dim, norm_fn, stride):
    return nn.Sequential(ResidualBlock(in_planes, dim, norm_fn, stride=stride), ResidualBlock(dim, dim, norm_fn, stride=1))


class BasicEncoder(nn.Module):
    """7x7 stem, three residual stages (64, 96, 128), 1x1 projection.  ``downsample`` = 3: 1/8
    resolution (RAFT); 2: 1/4 (RAFT-Stereo with n_downsample=2, whose stem keeps full resolution)."""

    def __init__(self, output_dim=128, norm_fn="batch", downsample=3):
        super().__init__()
        self.norm1 = _norm(norm_fn, 64)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=1 + (downsample > 2), padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = _stage(64, 64, norm_fn, 1)
        self.layer2 = _stage(64, 96, norm_fn, 1 + (downsample > 1))
        self.layer3 = _stage(96, 128, norm_fn, 1 + (downsample > 0))
        self.conv2 = nn.Conv2d(128, output_dim, kernel_size=1)

    def forward(self, x):
        x = self.relu1(self.norm1(self.conv1(x)))
        return self.conv2(self.layer3(self.layer2(self.layer1(x))))


class MultiBasicEncoder(nn.Module):
    """RAFT-Stereo's context encoder: the same trunk, two more stride-2 stages, and per GRU
    resolution one (hidden, context) pair of heads."""

    def __init__(self, output_dim, norm_fn="batch", downsample=3):
        super().__init__()
        self.norm1 = _norm(norm_fn, 64)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=1 + (downsample > 2), padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = _stage(64, 64, norm_fn, 1)
        self.layer2 = _stage(64, 96, norm_fn, 1 + (downsample > 1))
        self.layer3 = _stage(96, 128, norm_fn, 1 + (downsample > 0))
        self.layer4 = _stage(128, 128, norm_fn, 2)
        self.layer5 = _stage(128, 128, norm_fn, 2)
        head = lambda d: nn.Sequential(ResidualBlock(128, 128, norm_fn, stride=1), nn.Conv2d(128, d, 3, padding=1))
        self.outputs08 = nn.ModuleList([head(dim[2]) for dim in output_dim])
        self.outputs16 = nn.ModuleList([head(dim[1]) for dim in output_dim])
        self.outputs32 = nn.ModuleList([nn.Conv2d(128, dim[0], 3, padding=1) for dim in output_dim])

    def forward(self, x, num_layers=3):
        x = self.relu1(self.norm1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(x)))
        out08 = [f(x) for f in self.outputs08]
        if num_layers == 1:
            return (out08,)
        y = self.layer4(x)
        out16 = [f(y) for f in self.outputs16]
        if num_layers == 2:
            return (out08, out16)
        z = self.layer5(y)
        return (out08, out16, [f(z) for f in self.outputs32])


# -------------------------------------------------------------- update blocks

class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256, output_dim=2):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, output_dim, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.conv2(self.relu(self.conv1(x)))


class SepConvGRU(nn.Module):
    """Convolutional GRU whose gates are 1x5 convolutions, then a second step with 5x1 ones."""

    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        cat = hidden_dim + input_dim
        for tag, ks, pad in (("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))):
            for gate in "zrq":
                setattr(self, f"conv{gate}{tag}", nn.Conv2d(cat, hidden_dim, ks, padding=pad))

    def forward(self, h, x):
        for tag in "12":
            hx = torch.cat([h, x], dim=1)
            z = torch.sigmoid(getattr(self, "convz" + tag)(hx))
            r = torch.sigmoid(getattr(self, "convr" + tag)(hx))
            q = torch.tanh(getattr(self, "convq" + tag)(torch.cat([r * h, x], dim=1)))
            h = (1 - z) * h + z * q
        return h


class ConvGRU(nn.Module):
    """3x3 convolutional GRU; (cz, cr, cq) are the context's precomputed gate terms."""

    def __init__(self, hidden_dim, input_dim, kernel_size=3):
        super().__init__()
        p = kernel_size // 2
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=p)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=p)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=p)

    def forward(self, h, cz, cr, cq, *x_list):
        x = torch.cat(x_list, dim=1)
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(self.convz(hx) + cz)
        r = torch.sigmoid(self.convr(hx) + cr)
        q = torch.tanh(self.convq(torch.cat([r * h, x], dim=1)) + cq)
        return (1 - z) * h + z * q


class MotionEncoder(nn.Module):
    """Correlation window + current flow -> 126 motion features, the flow appended as channels 126-127.
    ``widths`` = (convc1, convc2, convf1, convf2) output channels: (256,192,128,64) in RAFT,
    (64,64,64,64) in RAFT-Stereo."""

    def __init__(self, cor_planes, widths):
        super().__init__()
        c1, c2, f1, f2 = widths
        self.convc1 = nn.Conv2d(cor_planes, c1, 1, padding=0)
        self.convc2 = nn.Conv2d(c1, c2, 3, padding=1)
        self.convf1 = nn.Conv2d(2, f1, 7, padding=3)
        self.convf2 = nn.Conv2d(f1, f2, 3, padding=1)
        self.conv = nn.Conv2d(c2 + f2, 128 - 2, 3, padding=1)

    def forward(self, flow, corr):
        cor = F.relu(self.convc2(F.relu(self.convc1(corr))))
        flo = F.relu(self.convf2(F.relu(self.convf1(flow))))
        out = F.relu(self.conv(torch.cat([cor, flo], dim=1)))
        return torch.cat([out, flow], dim=1)


class BasicUpdateBlock(nn.Module):
    """RAFT: motion encoder -> separable GRU -> flow head; the mask head's output is scaled by 0.25."""

    def __init__(self, corr_levels=4, corr_radius=4, hidden_dim=128):
        super().__init__()
        self.encoder = MotionEncoder(corr_levels * (2 * corr_radius + 1) ** 2, (256, 192, 128, 64))
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, 64 * 9, 1, padding=0))

    def forward(self, net, inp, corr, flow):
        motion = self.encoder(flow, corr)
        net = self.gru(net, torch.cat([inp, motion], dim=1))
        return net, 0.25 * self.mask(net), self.flow_head(net)


def pool2x(x):
    return F.avg_pool2d(x, 3, stride=2, padding=1)


def interp(x, dest):
    return F.interpolate(x, dest.shape[2:], mode="bilinear", align_corners=True)


class BasicMultiUpdateBlock(nn.Module):
    """RAFT-Stereo: up to three ConvGRUs (coarsest first) that exchange pooled / interpolated state;
    the finest one is fed by the motion encoder."""

    def __init__(self, hidden_dims, n_gru_layers, n_downsample, corr_levels=4, corr_radius=4):
        super().__init__()
        self.n_gru_layers = n_gru_layers
        self.encoder = MotionEncoder(corr_levels * (2 * corr_radius + 1), (64, 64, 64, 64))
        self.gru08 = ConvGRU(hidden_dims[2], 128 + hidden_dims[1] * (n_gru_layers > 1))
        self.gru16 = ConvGRU(hidden_dims[1], hidden_dims[0] * (n_gru_layers == 3) + hidden_dims[2])
        self.gru32 = ConvGRU(hidden_dims[0], hidden_dims[1])
        self.flow_head = FlowHead(hidden_dims[2], hidden_dim=256, output_dim=2)
        factor = 2 ** n_downsample
        self.mask = nn.Sequential(nn.Conv2d(hidden_dims[2], 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, (factor ** 2) * 9, 1, padding=0))

    def forward(self, net, inp, corr=None, flow=None, iter08=True, iter16=True, iter32=True, update=True):
        if iter32:
            net[2] = self.gru32(net[2], *(inp[2]), pool2x(net[1]))
        if iter16:
            if self.n_gru_layers > 2:
                net[1] = self.gru16(net[1], *(inp[1]), pool2x(net[0]), interp(net[2], net[1]))
            else:
                net[1] = self.gru16(net[1], *(inp[1]), pool2x(net[0]))
        if iter08:
            motion = self.encoder(flow, corr)
            if self.n_gru_layers > 1:
                net[0] = self.gru08(net[0], *(inp[0]), motion, interp(net[1], net[0]))
            else:
                net[0] = self.gru08(net[0], *(inp[0]), motion)
        if not update:
            return net
        return net, 0.25 * self.mask(net[0]), self.flow_head(net[0])


# ------------------------------------------------------------ whole networks

class RAFT(nn.Module):
    """RAFT optical flow, inference forward (the published ``test_mode``): returns
    (flow at 1/8 resolution, flow at full resolution) after ``iters`` refinement steps."""

    def __init__(self, corr_levels=4, corr_radius=4, hidden_dim=128, context_dim=128):
        super().__init__()
        self.corr_levels, self.corr_radius = corr_levels, corr_radius
        self.hidden_dim, self.context_dim = hidden_dim, context_dim
        self.fnet = BasicEncoder(output_dim=256, norm_fn="instance", downsample=3)
        self.cnet = BasicEncoder(output_dim=hidden_dim + context_dim, norm_fn="batch", downsample=3)
        self.update_block = BasicUpdateBlock(corr_levels, corr_radius, hidden_dim=hidden_dim)

    @torch.no_grad()
    def forward(self, image1, image2, iters=12, flow_init=None):
        image1 = 2 * (image1 / 255.0) - 1.0
        image2 = 2 * (image2 / 255.0) - 1.0
        fmap1, fmap2 = self.fnet(torch.cat([image1, image2], 0)).float().chunk(2, dim=0)
        pyr = R.corr_pyramid_2d(R.corr_volume_2d(fmap1, fmap2), self.corr_levels)
        net, inp = torch.split(self.cnet(image1), [self.hidden_dim, self.context_dim], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        B, _, H, W = image1.shape
        coords0 = R.coords_grid(B, H // 8, W // 8, image1.device)
        coords1 = coords0.clone()
        if flow_init is not None:
            coords1 = coords1 + flow_init
        mask = None
        for _ in range(iters):
            window = R.corr_lookup_2d(pyr, coords1, self.corr_radius)
            net, mask, delta = self.update_block(net, inp, window, coords1 - coords0)
            coords1 = coords1 + delta
        return coords1 - coords0, R.convex_upsample(coords1 - coords0, mask, 8)


class RAFTStereo(nn.Module):
    """RAFT-Stereo with the materialised per-row volume, inference forward: returns (x-flow field at
    1/2^n_downsample resolution (B,2,h,w), disparity at full resolution (B,1,H,W), negative for a
    rectified left/right pair)."""

    def __init__(self, hidden_dims=(128, 128, 128), n_gru_layers=3, n_downsample=2, corr_levels=4, corr_radius=4,
                 context_norm="batch", slow_fast_gru=False):
        super().__init__()
        hidden_dims = list(hidden_dims)
        self.hidden_dims, self.n_gru_layers, self.n_downsample = hidden_dims, n_gru_layers, n_downsample
        self.corr_levels, self.corr_radius, self.slow_fast_gru = corr_levels, corr_radius, slow_fast_gru
        self.cnet = MultiBasicEncoder(output_dim=[hidden_dims, hidden_dims], norm_fn=context_norm, downsample=n_downsample)
        self.update_block = BasicMultiUpdateBlock(hidden_dims, n_gru_layers, n_downsample, corr_levels, corr_radius)
        self.context_zqr_convs = nn.ModuleList(
            [nn.Conv2d(hidden_dims[i], hidden_dims[i] * 3, 3, padding=1) for i in range(n_gru_layers)])
        self.fnet = BasicEncoder(output_dim=256, norm_fn="instance", downsample=n_downsample)

    @torch.no_grad()
    def forward(self, image1, image2, iters=12, flow_init=None):
        n = self.n_gru_layers
        image1 = 2 * (image1 / 255.0) - 1.0
        image2 = 2 * (image2 / 255.0) - 1.0
        cnet_list = self.cnet(image1, num_layers=n)
        fmap1, fmap2 = self.fnet(torch.cat([image1, image2], 0)).float().chunk(2, dim=0)
        net_list = [torch.tanh(x[0]) for x in cnet_list]
        inp_list = [torch.relu(x[1]) for x in cnet_list]
        inp_list = [list(conv(i).split(split_size=conv.out_channels // 3, dim=1))
                    for i, conv in zip(inp_list, self.context_zqr_convs)]
        pyr = R.corr_pyramid_1d(R.corr_volume_1d(fmap1, fmap2), self.corr_levels)
        B, _, h, w = net_list[0].shape
        coords0 = R.coords_grid(B, h, w, image1.device)
        coords1 = coords0.clone()
        if flow_init is not None:
            coords1 = coords1 + flow_init
        mask = None
        for _ in range(iters):
            window = R.corr_lookup_1d(pyr, coords1, self.corr_radius)
            flow = coords1 - coords0
            if n == 3 and self.slow_fast_gru:
                net_list = self.update_block(net_list, inp_list, iter32=True, iter16=False, iter08=False, update=False)
            if n >= 2 and self.slow_fast_gru:
                net_list = self.update_block(net_list, inp_list, iter32=n == 3, iter16=True, iter08=False, update=False)
            net_list, mask, delta = self.update_block(net_list, inp_list, window, flow, iter32=n == 3, iter16=n >= 2)
            delta[:, 1] = 0.0                                    # epipolar: no y motion
            coords1 = coords1 + delta
        up = R.convex_upsample(coords1 - coords0, mask, 2 ** self.n_downsample)
        return coords1 - coords0, up[:, :1]


# --------------------------------------------------------------------- twins

def flow_twin(model, device="cpu") -> RAFT:
    """A RAFT of this module with the parameters of ``model`` (anything with ``cfg.corr_levels /
    corr_radius / hidden_dim / context_dim`` and a ``state_dict()`` under the published names)."""
    c = model.cfg
    twin = RAFT(c.corr_levels, c.corr_radius, c.hidden_dim, c.context_dim).to(device).eval()
    twin.load_state_dict({k: v.detach().to(device) for k, v in model.state_dict().items()}, strict=True)
    return twin


def stereo_twin(model, device="cpu") -> RAFTStereo:
    """A RAFTStereo of this module with the parameters of ``model`` (``cfg.hidden_dims / n_gru_layers /
    n_downsample / corr_levels / corr_radius / context_norm / slow_fast_gru`` + ``state_dict()``)."""
    c = model.cfg
    twin = RAFTStereo(c.hidden_dims, c.n_gru_layers, c.n_downsample, c.corr_levels, c.corr_radius, c.context_norm,
                      c.slow_fast_gru).to(device).eval()
    twin.load_state_dict({k: v.detach().to(device) for k, v in model.state_dict().items()}, strict=True)
    return twin
