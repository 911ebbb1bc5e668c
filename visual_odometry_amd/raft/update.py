"""Recurrent update operators of RAFT and RAFT-Stereo.

``nn.Module`` trees for the parameters; inside a product forward pass a GRU step is two hand-written
convolution kernels with the gate arithmetic in their stores, the motion encoders and heads run on the
same kernel family (raft/layers.py), and the stock PyTorch operators below are what runs with the fused
passes off.  Sub-module names follow the published models (``encoder.convc1``, ``gru.convz1``,
``flow_head.conv1``, ``mask.0`` ...) so trained state dicts load unchanged.  The
only departure in behaviour is an optimisation that cannot change the result:
the up-sampling mask head is evaluated only when the caller asks for it (the
last refinement iteration in inference), and it is returned UN-scaled -- the
0.25 factor is folded into the HIP convex-upsampling kernel.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .layers import conv2d_nobias, conv_act, conv_plain, fused_on, padded_input_twin, split_on, split_pieces, weight_for


def _merged_zr_weight(owner: nn.Module, tag: str, convz: nn.Conv2d, convr: nn.Conv2d, dtype) -> torch.Tensor:
    """cat(convz.weight, convr.weight) in ``dtype``: z and r read the same input, so one
    convolution with 2C output channels replaces two.  Cached; rebuilt if a weight
    changes or moves."""
    key = (convz.weight._version, convz.weight.data_ptr(), convr.weight._version, convr.weight.data_ptr(), dtype)
    cached = getattr(owner, "_vo_zr_" + tag, None)
    if cached is None or cached[0] != key:
        cached = (key, torch.cat([convz.weight, convr.weight], dim=0).detach().to(dtype).contiguous())
        setattr(owner, "_vo_zr_" + tag, cached)
    return cached[1]


def _split_gru_operands(owner, tag, convz, convr, convq, fold=None):
    """Split-bf16 operands of a GRU step's two convolutions (merged z|r weights and biases, q weights), cached
    on the module; rebuilt if a weight or bias changes or moves.  ``fold`` = (lo, hi): the input channels
    lo..hi-1 carry a tensor that does not change over the refinement iterations; they are cut out of the
    step's weights and returned as separate (z|r, q) weights, whose convolution the caller evaluates once."""
    tensors = [convz.weight, convr.weight, convq.weight] + [c.bias for c in (convz, convr, convq) if c.bias is not None]
    pieces = split_pieces()
    key = (fold, pieces) + tuple((t._version, t.data_ptr()) for t in tensors)
    table = getattr(owner, "_vo_gru_split_" + tag, None)
    if table is None:
        table = {}
        setattr(owner, "_vo_gru_split_" + tag, table)
    cached = table.get(pieces)
    if cached is None or cached[0] != key:
        with torch.no_grad():
            wzr = torch.cat([convz.weight, convr.weight], dim=0).float()
            wq = convq.weight.float()
            folded = None
            if fold is not None:
                lo, hi = fold
                folded = (wzr[:, lo:hi].contiguous(), wq[:, lo:hi].contiguous())
                wzr = torch.cat([wzr[:, :lo], wzr[:, hi:]], dim=1)
                wq = torch.cat([wq[:, :lo], wq[:, hi:]], dim=1)
            Uzr = ops.gru_split_filter(wzr.contiguous(), pieces)
            Uq = ops.gru_split_filter(wq.contiguous(), pieces)
            bzr = None
            if convz.bias is not None and convr.bias is not None:
                bzr = torch.cat([convz.bias, convr.bias]).float().contiguous()
        cached = (key, Uzr, bzr, Uq, folded)
        table[pieces] = cached
    return cached[1], cached[2], cached[3], cached[4]


def _split_gru_eligible(h: torch.Tensor, in_channels: int, convz: nn.Conv2d, convr: nn.Conv2d) -> bool:
    """True when a GRU step over [h | x] with ``in_channels`` channels can run as two split-bf16 convolutions
    with the gate arithmetic in their stores."""
    ks = tuple(convz.kernel_size)
    B, C, H, W = h.shape
    return (split_on() and h.dtype == torch.float32 and h.is_cuda and h.is_contiguous() and C % 64 == 0
            and in_channels % 16 == 0 and W % 4 == 0 and ks in ((3, 3), (1, 5), (5, 1))
            and tuple(convz.padding) == (ks[0] // 2, ks[1] // 2) and (convz.bias is None) == (convr.bias is None)
            and in_channels * H * W * 4 < 2 ** 31 - 1)


def _fused_gru_step(owner, tag, buf, h, C, convz, convr, convq, ctx=(None, None, None), h_copy=None, fold=None):
    """One GRU step on buf = [h | x] (B, C+X, H, W); returns the new hidden state (and leaves it in ``h_copy``).
    Split-bf16 arithmetic: two convolutions with the gate arithmetic in their stores (buf is only read, r*h
    goes to a tensor of its own which the q convolution reads in place of buf's first C channels).  Otherwise:
    two bias-free convolutions and two pointwise passes, buf's first C channels overwritten with r*h in between.
    ``fold`` = (lo, hi): buf does NOT hold the model's input channels lo..hi-1; their contribution arrives in
    ``ctx`` (see SepConvGRU.constant_terms) -- split-bf16 path only, the caller checked _split_gru_eligible."""
    ks = tuple(convz.kernel_size)
    if fold is not None or (buf.dtype == torch.float32 and buf.is_contiguous() and _split_gru_eligible(h, buf.shape[1], convz, convr)):
        Uzr, bzr, Uq, _ = _split_gru_operands(owner, tag, convz, convr, convq, fold)
        z, rh = ops.gru_conv_zr(buf, Uzr, bzr, ctx[0], ctx[1], h, ks)
        return ops.gru_conv_q(buf, rh, Uq, convq.bias, ctx[2], z, h, ks, h_copy)
    zr = conv2d_nobias(buf, _merged_zr_weight(owner, tag, convz, convr, buf.dtype), (1, 1), convz.padding)
    z = ops.gru_gates(zr, convz.bias, convr.bias, ctx[0], ctx[1], h, buf[:, :C])
    q = conv2d_nobias(buf, weight_for(convq, buf), (1, 1), convq.padding)
    return ops.gru_update(q, convq.bias, ctx[2], z, h, torch.empty_like(h), h_copy)


class FlowHead(nn.Module):
    def __init__(self, input_dim: int = 128, hidden_dim: int = 256, output_dim: int = 2):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, output_dim, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return conv_plain(conv_act(x, self.conv1), self.conv2)


class SepConvGRU(nn.Module):
    """GRU with separable 1x5 then 5x1 gates (RAFT-Flow)."""

    def __init__(self, hidden_dim: int = 128, input_dim: int = 192 + 128):
        super().__init__()
        cat = hidden_dim + input_dim
        self.convz1 = nn.Conv2d(cat, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(cat, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(cat, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(cat, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(cat, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(cat, hidden_dim, (5, 1), padding=(2, 0))

    @staticmethod
    def _step(h, x, convz, convr, convq):
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(convz(hx))
        r = torch.sigmoid(convr(hx))
        q = torch.tanh(convq(torch.cat([r * h, x], dim=1)))
        return (1 - z) * h + z * q

    def constant_terms(self, const: torch.Tensor, lo: int):
        """The part of the six gate convolutions that reads input channels lo..lo+K-1 = ``const`` (B,K,H,W), a
        tensor that stays the same over all refinement iterations (RAFT's context features): evaluated once,
        handed to every step as additive terms -- the per-iteration convolutions then run over a third fewer
        input channels.  Returns ((cz1, cr1, cq1), (cz2, cr2, cq2)), channel slices of two tensors."""
        C = self.convz1.out_channels
        out = []
        for tag, cz, cr, cq in (("1", self.convz1, self.convr1, self.convq1), ("2", self.convz2, self.convr2, self.convq2)):
            _, _, _, folded = _split_gru_operands(self, tag, cz, cr, cq, (lo, lo + const.shape[1]))
            both = torch.cat([conv2d_nobias(const, folded[0], (1, 1), cz.padding), conv2d_nobias(const, folded[1], (1, 1), cz.padding)], dim=1)
            out.append(tuple(both.split(C, dim=1)))
        return tuple(out)

    def forward(self, h, x, buf=None, ctx=None, fold=None):
        """``x`` may be a tensor or a list of tensors that together form the input.  ``buf``
        (fused path): a (B, C+X, H, W) buffer whose channels C.. already hold the input and
        whose first C channels hold ``h``; it is left holding the NEW hidden state in its
        first C channels, ready for the next iteration.  ``ctx`` / ``fold``: the terms of
        ``constant_terms`` and the channel range they stand for (buf then holds the other inputs only)."""
        if fused_on() and h.is_cuda and h.dtype in (torch.float32, torch.float16):
            C = h.shape[1]
            if buf is None:
                parts = list(x) if isinstance(x, (list, tuple)) else [x]
                buf = torch.cat([h] + parts, dim=1)                      # the only copy of the step
            c1, c2 = ctx if ctx is not None else ((None, None, None), (None, None, None))
            h = _fused_gru_step(self, "1", buf, h, C, self.convz1, self.convr1, self.convq1, c1, h_copy=buf[:, :C], fold=fold)
            return _fused_gru_step(self, "2", buf, h, C, self.convz2, self.convr2, self.convq2, c2, h_copy=buf[:, :C], fold=fold)
        if isinstance(x, (list, tuple)):
            x = torch.cat(list(x), dim=1)
        h = self._step(h, x, self.convz1, self.convr1, self.convq1)
        return self._step(h, x, self.convz2, self.convr2, self.convq2)


class ConvGRU(nn.Module):
    """3x3 GRU whose context contribution (cz, cr, cq) is precomputed once per
    image (RAFT-Stereo)."""

    def __init__(self, hidden_dim: int, input_dim: int, kernel_size: int = 3):
        super().__init__()
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=kernel_size // 2)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=kernel_size // 2)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=kernel_size // 2)

    def forward(self, h, cz, cr, cq, *x_list, buf=None):
        """``buf`` (fused path): prepared (B, C+X, H, W) buffer as in SepConvGRU.forward."""
        if fused_on() and h.is_cuda and h.dtype in (torch.float32, torch.float16):
            C = h.shape[1]
            if buf is None:
                buf = torch.cat([h] + list(x_list), dim=1)
            return _fused_gru_step(self, "0", buf, h, C, self.convz, self.convr, self.convq, (cz, cr, cq),
                                   h_copy=buf[:, :C])
        x = torch.cat(x_list, dim=1)
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(self.convz(hx) + cz)
        r = torch.sigmoid(self.convr(hx) + cr)
        q = torch.tanh(self.convq(torch.cat([r * h, x], dim=1)) + cq)
        return (1 - z) * h + z * q


class FlowMotionEncoder(nn.Module):
    """Correlation window + current flow -> 128 motion features (RAFT-Flow widths)."""

    def __init__(self, cor_planes: int):
        super().__init__()
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)

    def forward(self, flow, corr, out=None):
        """``out`` (fused path): channel slice (B,126,H,W) of the GRU's input buffer that
        receives the motion features."""
        if fused_on():
            B, _, H, W = flow.shape
            both = torch.empty((B, 192 + 64, H, W), device=flow.device, dtype=flow.dtype)    # [cor | flo], written in place
            # corr may carry zero planes behind the window (padded to a multiple of 16 channels)
            conv_act(conv_act(corr, padded_input_twin(self.convc1, corr.shape[1])), self.convc2, out=both[:, :192])
            conv_act(conv_act(flow, self.convf1), self.convf2, out=both[:, 192:])
            return [conv_act(both, self.conv, out=out), flow]      # the GRU concatenates once, with its other inputs
        cor = conv_act(conv_act(corr[:, :self.convc1.in_channels], self.convc1), self.convc2)
        flo = conv_act(conv_act(flow, self.convf1), self.convf2)
        out = conv_act(torch.cat([cor, flo], dim=1), self.conv)
        return torch.cat([out, flow], dim=1)


class StereoMotionEncoder(nn.Module):
    """Same structure with RAFT-Stereo's widths (64-channel branches)."""

    def __init__(self, cor_planes: int):
        super().__init__()
        self.convc1 = nn.Conv2d(cor_planes, 64, 1, padding=0)
        self.convc2 = nn.Conv2d(64, 64, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 64, 7, padding=3)
        self.convf2 = nn.Conv2d(64, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 64, 128 - 2, 3, padding=1)

    def forward(self, flow, corr, out=None):
        """``out`` (fused path): channel slice (B,126,H,W) of the GRU's input buffer that
        receives the motion features."""
        if fused_on():
            B, _, H, W = flow.shape
            both = torch.empty((B, 64 + 64, H, W), device=flow.device, dtype=flow.dtype)    # [cor | flo], written in place
            conv_act(conv_act(corr, self.convc1), self.convc2, out=both[:, :64])
            conv_act(conv_act(flow, self.convf1), self.convf2, out=both[:, 64:])
            return [conv_act(both, self.conv, out=out), flow]      # the GRU concatenates once, with its other inputs
        cor = conv_act(conv_act(corr, self.convc1), self.convc2)
        flo = conv_act(conv_act(flow, self.convf1), self.convf2)
        out = conv_act(torch.cat([cor, flo], dim=1), self.conv)
        return torch.cat([out, flow], dim=1)


class BasicUpdateBlock(nn.Module):
    """RAFT-Flow update: motion encoder -> SepConvGRU -> flow head (+ mask head)."""

    def __init__(self, corr_levels: int = 4, corr_radius: int = 4, hidden_dim: int = 128):
        super().__init__()
        self.encoder = FlowMotionEncoder(corr_levels * (2 * corr_radius + 1) ** 2)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, 64 * 9, 1, padding=0))

    def forward(self, net, inp, corr, flow, want_mask: bool = True, state=None):
        """``state`` (fused path): a dict the caller keeps across the iterations of one forward
        pass; it holds the GRU's input buffer [h | inp | motion | flow], in which only the
        motion and flow channels are rewritten per iteration (h is left there by the GRU)."""
        if fused_on() and state is not None and net.is_cuda:
            C, X = net.shape[1], inp.shape[1]
            buf = state.get("buf")
            if buf is None:
                B, _, H, W = net.shape
                # The context features do not change over the iterations: where the GRU runs on the split-bf16
                # kernels their share of the gate convolutions is evaluated once (state["ctx"]) and the buffer
                # holds [h | motion | flow] only; otherwise [h | context | motion | flow].
                fold = (C, C + X) if (_split_gru_eligible(net.contiguous(), C + 128, self.gru.convz1, self.gru.convr1)
                                      and inp.dtype == torch.float32) else None
                state["fold"] = fold
                keep = 0 if fold is not None else X
                buf = torch.empty((B, C + keep + 128, H, W), device=net.device, dtype=net.dtype)
                buf[:, :C].copy_(net)
                if fold is None:
                    buf[:, C:C + X].copy_(inp)
                else:
                    state["ctx"] = self.gru.constant_terms(inp.contiguous(), C)
                state["buf"] = buf
            fold = state["fold"]
            m0 = C + (0 if fold is not None else X)
            self.encoder(flow, corr, out=buf[:, m0:m0 + 126])
            buf[:, m0 + 126:].copy_(flow)
            net = self.gru(net.contiguous(), None, buf=buf, ctx=state.get("ctx"), fold=fold)
        else:
            motion = self.encoder(flow, corr)
            if isinstance(motion, list):
                net = self.gru(net, [inp] + motion)
            else:
                net = self.gru(net, torch.cat([inp, motion], dim=1))
        delta_flow = self.flow_head(net)
        raw_mask = conv_plain(conv_act(net, self.mask[0]), self.mask[2]) if want_mask else None   # caller scales by 0.25
        return net, raw_mask, delta_flow


def pool2x(x):
    if fused_on() and x.is_cuda and x.dtype in (torch.float32, torch.float16) and x.is_contiguous():
        return ops.avg_pool3x3s2(x)
    return F.avg_pool2d(x, 3, stride=2, padding=1)


def interp(x, dest):
    if fused_on() and x.is_cuda and x.dtype in (torch.float32, torch.float16) and x.is_contiguous():
        return ops.resize_bilinear(x, dest.shape[2:])
    return F.interpolate(x, dest.shape[2:], mode="bilinear", align_corners=True)


class BasicMultiUpdateBlock(nn.Module):
    """RAFT-Stereo update: up to three ConvGRUs at 1/32-1/16-1/8 of the working
    resolution exchanging pooled / interpolated state, finest one fed by the
    motion encoder."""

    def __init__(self, hidden_dims, n_gru_layers: int, n_downsample: int, corr_levels: int = 4,
                 corr_radius: int = 4):
        super().__init__()
        self.n_gru_layers = n_gru_layers
        self.encoder = StereoMotionEncoder(corr_levels * (2 * corr_radius + 1))
        encoder_output_dim = 128
        self.gru08 = ConvGRU(hidden_dims[2], encoder_output_dim + hidden_dims[1] * (n_gru_layers > 1))
        self.gru16 = ConvGRU(hidden_dims[1], hidden_dims[0] * (n_gru_layers == 3) + hidden_dims[2])
        self.gru32 = ConvGRU(hidden_dims[0], hidden_dims[1])
        self.flow_head = FlowHead(hidden_dims[2], hidden_dim=256, output_dim=2)
        factor = 2 ** n_downsample
        self.mask = nn.Sequential(nn.Conv2d(hidden_dims[2], 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, (factor ** 2) * 9, 1, padding=0))

    def forward(self, net, inp, corr=None, flow=None, iter08=True, iter16=True, iter32=True, update=True,
                want_mask: bool = True, state=None):
        # fused path: every level keeps a persistent input buffer [h | the other levels' state] across the
        # iterations of one forward pass; the pooled / interpolated states are written into their channel
        # slices, h is left in the first channels by the GRU of the previous call
        buffers = fused_on() and state is not None and net[0].is_cuda and net[0].dtype in (torch.float32, torch.float16)

        def level_buffer(key, h, extra):
            buf = state.get(key)
            if buf is None:
                B, C, H, W = h.shape
                buf = torch.empty((B, C + extra, H, W), device=h.device, dtype=h.dtype)
                buf[:, :C].copy_(h)
                state[key] = buf
            return buf

        if iter32 and buffers:
            C = net[2].shape[1]
            buf = level_buffer("buf32", net[2], net[1].shape[1])
            ops.avg_pool3x3s2(net[1].contiguous(), out=buf[:, C:])
            net[2] = self.gru32(net[2], *(inp[2]), buf=buf)
        elif iter32:
            net[2] = self.gru32(net[2], *(inp[2]), pool2x(net[1]))
        if iter16 and buffers:
            C, C0 = net[1].shape[1], net[0].shape[1]
            buf = level_buffer("buf16", net[1], C0 + (net[2].shape[1] if self.n_gru_layers > 2 else 0))
            ops.avg_pool3x3s2(net[0].contiguous(), out=buf[:, C:C + C0])
            if self.n_gru_layers > 2:
                ops.resize_bilinear(net[2].contiguous(), net[1].shape[2:], out=buf[:, C + C0:])
            net[1] = self.gru16(net[1], *(inp[1]), buf=buf)
        elif iter16:
            if self.n_gru_layers > 2:
                net[1] = self.gru16(net[1], *(inp[1]), pool2x(net[0]), interp(net[2], net[1]))
            else:
                net[1] = self.gru16(net[1], *(inp[1]), pool2x(net[0]))
        if iter08 and fused_on() and state is not None and net[0].is_cuda:
            # persistent input buffer [h | motion(126) | flow(2) | interp(net[1])]: producers write
            # their channel slices in place, h is left there by the GRU of the previous iteration
            C = net[0].shape[1]
            buf = state.get("buf08")
            if buf is None:
                B, _, H, W = net[0].shape
                extra = net[1].shape[1] if self.n_gru_layers > 1 else 0
                buf = torch.empty((B, C + 128 + extra, H, W), device=net[0].device, dtype=net[0].dtype)
                buf[:, :C].copy_(net[0])
                state["buf08"] = buf
            self.encoder(flow, corr, out=buf[:, C:C + 126])
            buf[:, C + 126:C + 128].copy_(flow)
            if self.n_gru_layers > 1:
                ops.resize_bilinear(net[1], net[0].shape[2:], out=buf[:, C + 128:])
            net[0] = self.gru08(net[0], *(inp[0]), buf=buf)
        elif iter08:
            motion = self.encoder(flow, corr)
            motion = motion if isinstance(motion, list) else [motion]
            if self.n_gru_layers > 1:
                net[0] = self.gru08(net[0], *(inp[0]), *motion, interp(net[1], net[0]))
            else:
                net[0] = self.gru08(net[0], *(inp[0]), *motion)
        if not update:
            return net
        delta_flow = self.flow_head(net[0])
        raw_mask = conv_plain(conv_act(net[0], self.mask[0]), self.mask[2]) if want_mask else None   # caller scales by 0.25
        return net, raw_mask, delta_flow
