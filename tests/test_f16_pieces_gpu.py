"""The split kernels with float16 pieces (VO_PIECES_F16): each float32 operand cut into two float16 parts, three
matrix-core terms per product, exact power-of-two scaling for float16's exponent range.

The reference runs float32 throughout (mixed_precision=False, flow/flow_map_RAFT.py:38,
stereo/stereo_disparity_RAFT.py:43); this arithmetic stands for it, so its bar is the float32 matrix
instruction's own error: every layer type is checked against the float64 convolution with F32_GRADE_TOL of the
largest output AND against the float32-instruction kernel of the same layer on the same data (the split result
may not be further from float64 than twice that kernel's result plus a rounding of the output).
"""
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu
DEV = "cuda"
F32_GRADE_TOL = 2.5e-6             # of the largest output; the float32 matrix instruction measures 0.7-1.7e-6
BF16_FLOOR = 2.0e-6                # bf16 pieces cannot meet the bar above: the test would notice a mix-up of formats


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _err(got, want):
    return (got.double() - want).abs().max().item() / max(1e-30, want.abs().max().item())


@pytest.mark.parametrize("N,C,H,W,K", [(2, 16, 16, 32, 64), (1, 16, 37, 54, 40), (2, 64, 60, 80, 126), (3, 128, 33, 46, 192),
                                       (1, 16, 1, 2, 33), (1, 256, 15, 20, 128), (2, 64, 120, 160, 64), (1, 384, 30, 40, 256)])
def test_conv3x3_float16_pieces_is_float32_grade(vo, N, C, H, W, K):
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=50).to(DEV)
    w = (_rand(K, C, 3, 3, seed=51) / (3 * C ** 0.5)).to(DEV)
    want = Fn.conv2d(x.double(), w.double(), None, 1, 1)
    U16, Ub = ops.conv3x3_split_filter(w, "f16"), ops.conv3x3_split_filter(w, "bf16")
    assert U16.vo_pieces == "f16" and 2 ** 13 <= float(w.abs().max()) * 2.0 ** U16.vo_wexp < 2 ** 14
    got = ops.conv3x3_winograd(x, U16, K)
    e16, eb = _err(got, want), _err(ops.conv3x3_winograd(x, Ub, K), want)
    assert e16 < F32_GRADE_TOL, e16
    assert torch.equal(got, ops.conv3x3_winograd(x, U16, K))                # run-to-run identical
    if C * 9 >= 576:
        assert eb > BF16_FLOOR and e16 < 0.5 * eb                           # the two formats are told apart
    if C % 8 == 0 and K >= 32 and C * 9 >= 576:
        # the kernel the float32-instruction path runs this layer on (Winograd F(4x4,3x3)) is further from float64
        e32 = _err(ops.conv3x3_winograd(x, ops.winograd_filter(w, 4), K), want)
        assert e16 < e32, (e16, e32)
    # epilogue, residual join, plane sums, normalise-on-load: same code as with bf16 pieces, after the unscaling
    scale, shift, skip = _rand(K, seed=72).to(DEV), _rand(K, seed=73).to(DEV), _rand(N, K, H, W, seed=74).to(DEV)
    b = lambda v: v[None, :, None, None]
    ref = torch.relu(skip.double() + torch.relu(want * b(scale).double() + b(shift).double()))
    assert _err(ops.conv3x3_winograd(x, U16, K, scale, shift, True, skip), ref) < 2 * F32_GRADE_TOL
    stats = torch.zeros((N, K, 2), device=DEV, dtype=torch.float64)
    raw = ops.conv3x3_winograd(x, U16, K, stats=stats)
    assert torch.equal(raw, got)
    want_stats = torch.stack([want.sum((2, 3)), (want * want).sum((2, 3))], dim=2)
    assert (stats - want_stats).abs().max().item() < 1e-5 * max(1.0, want_stats.abs().max().item())
    if C <= 128 and H * W > 1:
        st = torch.stack([x.double().sum((2, 3)), (x.double() ** 2).sum((2, 3))], dim=2).contiguous()
        gotn = ops.conv3x3_winograd(x, U16, K, in_stats=st, in_eps=1e-5)
        wantn = Fn.conv2d(torch.relu(Fn.instance_norm(x.double(), eps=1e-5)), w.double(), None, 1, 1)
        assert _err(gotn, wantn) < 2 * F32_GRADE_TOL


@pytest.mark.parametrize("x_scale,w_scale", [(1e-4, 1.0), (1.0, 1e-5), (3e-5, 3e3), (5e3, 1e-3), (1.0, 7e4), (1e-7, 1.0)])
def test_float16_pieces_keep_their_precision_over_the_exponent_range(vo, x_scale, w_scale):
    """Inputs and weights far from 1: the weights are brought into float16's range by an exact power of two
    on the host, an input's low part is kept 2^11 times too large; the result keeps the float32 grade for inputs
    anywhere between 6.1e-5 and 65504 in magnitude, whatever the weights' scale, and degrades
    only gradually under that (absolute error 2^-36 per element)."""
    from visual_odometry_amd import ops
    N, C, H, W, K = 2, 64, 30, 40, 64
    x = (_rand(N, C, H, W, seed=60) * x_scale).to(DEV)
    w = (_rand(K, C, 3, 3, seed=61) * w_scale / 24).to(DEV)
    want = Fn.conv2d(x.double(), w.double(), None, 1, 1)
    got = ops.conv3x3_winograd(x, ops.conv3x3_split_filter(w, "f16"), K)
    assert torch.isfinite(got).all()
    tol = F32_GRADE_TOL if x_scale >= 1e-5 else 2e-4          # 1e-7: every value under float16's normal range
    assert _err(got, want) < tol, _err(got, want)


def test_float16_pieces_inputs_across_many_binades_and_out_of_range(vo):
    from visual_odometry_amd import ops
    N, C, H, W, K = 1, 32, 24, 32, 64
    w = (_rand(K, C, 3, 3, seed=63) / 17).to(DEV)
    U = ops.conv3x3_split_filter(w, "f16")
    x = _rand(N, C, H, W, seed=62).to(DEV) * torch.exp2(torch.randint(-12, 12, (N, C, H, W), device=DEV).float())
    want = Fn.conv2d(x.double(), w.double(), None, 1, 1)
    assert _err(ops.conv3x3_winograd(x, U, K), want) < F32_GRADE_TOL
    # a value past float16's range does not come back as a wrong number: it poisons what it touches, nothing else
    x2 = _rand(N, C, H, W, seed=64).to(DEV)
    x2[0, 3, 10, 12] = 7.0e4
    got = ops.conv3x3_winograd(x2, U, K)
    assert torch.isnan(got[0, :, 9:12, 11:14]).all()
    clean = torch.ones_like(got, dtype=torch.bool); clean[0, :, 9:12, 11:14] = False
    want2 = Fn.conv2d(x2.double(), w.double(), None, 1, 1)
    assert torch.isfinite(got[clean]).all() and (got.double() - want2)[clean].abs().max().item() < F32_GRADE_TOL * want2[clean].abs().max().item()
    # 65504 itself is in range
    x2[0, 3, 10, 12] = 65504.0
    got = ops.conv3x3_winograd(x2, U, K)
    assert torch.isfinite(got).all() and _err(got, Fn.conv2d(x2.double(), w.double(), None, 1, 1)) < F32_GRADE_TOL
    # the two formats cannot be mixed up: arranged weights carry theirs, the C ABI refuses a stray exponent
    from visual_odometry_amd import _lib
    Ub = ops.conv3x3_split_filter(w, "bf16")
    assert Ub.vo_pieces == "bf16" and Ub.vo_wexp == 0
    lib = _lib.load()
    assert lib.vo_conv3x3_bf16x3_filter(w.data_ptr(), Ub.data_ptr(), K, C, _lib.VO_PIECES_BF16, 3, None) == _lib.VO_ERR_INVALID_ARG
    assert lib.vo_conv3x3_bf16x3_filter(w.data_ptr(), Ub.data_ptr(), K, C, 2, 0, None) == _lib.VO_ERR_INVALID_ARG


@pytest.mark.parametrize("vertical", [False, True])
@pytest.mark.parametrize("N,C,H,W,K", [(2, 32, 33, 22, 40), (2, 384, 30, 40, 256), (3, 384, 60, 80, 128), (1, 16, 1, 2, 33)])
def test_conv5_float16_pieces(vo, N, C, H, W, K, vertical):
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=110).to(DEV)
    w = (_rand(*((K, C, 5, 1) if vertical else (K, C, 1, 5)), seed=111) / (5 * C) ** 0.5).to(DEV)
    U = ops.conv5_split_filter(w, "f16")
    want = Fn.conv2d(x.double(), w.double(), None, 1, (2, 0) if vertical else (0, 2))
    got = ops.conv5_split(x, U, K, vertical)
    assert _err(got, want) < F32_GRADE_TOL
    if C % 8 == 0 and K >= 32:
        e32 = _err(ops.conv5_winograd(x, ops.winograd1d_filter(w), K, vertical), want)       # the float32 path's kernel: F(4,5)
        assert _err(got, want) < e32 + 2e-7
    bias = _rand(K, seed=112).to(DEV)
    wide = torch.full((N, 3 + K + 2, H, W), 1.75, device=DEV)
    ops.conv5_split(x, U, K, vertical, None, bias, True, out=wide[:, 3:3 + K])
    assert _err(wide[:, 3:3 + K], torch.relu(want + bias.double()[None, :, None, None])) < 2 * F32_GRADE_TOL
    assert torch.all(wide[:, :3] == 1.75) and torch.all(wide[:, 3 + K:] == 1.75)


@pytest.mark.parametrize("N,C,H,W,K", [(2, 64, 48, 64, 96), (1, 96, 30, 40, 128), (2, 64, 37, 24, 64), (3, 128, 120, 160, 128)])
def test_stride2_layers_float16_pieces(vo, N, C, H, W, K):
    """3x3 stride 2, 1x1 stride 2 and the two of them from one staged patch (bit for bit the single kernels)."""
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=180).to(DEV)
    w3 = (_rand(K, C, 3, 3, seed=181) / (3 * C ** 0.5)).to(DEV)
    w1 = (_rand(K, C, 1, 1, seed=182) * 37.0 / C ** 0.5).to(DEV)            # a different power of two than w3's
    A3, A1 = ops.conv3x3_split_filter(w3, "f16"), ops.conv1x1_split_filter(w1, "f16")
    assert A3.vo_wexp != A1.vo_wexp
    want3, want1 = Fn.conv2d(x.double(), w3.double(), None, 2, 1), Fn.conv2d(x.double(), w1.double(), None, 2, 0)
    y3, y1 = ops.conv3x3s2_split(x, A3, K), ops.conv1x1s2_split(x, A1, K)
    assert _err(y3, want3) < F32_GRADE_TOL and _err(y1, want1) < F32_GRADE_TOL
    st3, st1 = (torch.zeros((N, K, 2), device=DEV, dtype=torch.float64) for _ in range(2))
    y, sc = ops.conv3x3s2_shortcut_split(x, A3, A1, K, stats=st3, stats2=st1)
    assert torch.equal(y, y3) and torch.equal(sc, y1)
    for st, want in ((st3, want3), (st1, want1)):
        sums = torch.stack([want.sum((2, 3)), (want * want).sum((2, 3))], dim=2)
        assert (st - sums).abs().max().item() < 1e-5 * max(1.0, sums.abs().max().item())
    with pytest.raises(ValueError):
        ops.conv3x3s2_shortcut_split(x, A3, ops.conv1x1_split_filter(w1, "bf16"), K)       # formats must agree


@pytest.mark.parametrize("N,C,H,W,K,stride", [(2, 1, 96, 128, 64, 1), (1, 2, 60, 80, 128, 1), (3, 1, 61, 191, 64, 2), (2, 2, 33, 70, 40, 2)])
def test_conv7x7_float16_pieces(vo, N, C, H, W, K, stride):
    from visual_odometry_amd import ops
    x = (_rand(N, C, H, W, seed=130) * (40.0 if C == 2 else 1.0)).to(DEV)       # C = 2: a flow field in pixels
    w = (_rand(K, C, 7, 7, seed=131) / (7 * C ** 0.5)).to(DEV)
    A = ops.conv7x7_split_filter(w, "f16")
    want = Fn.conv2d(x.double(), w.double(), None, stride, 3)
    got = ops.conv7x7_split(x, A, K, stride=stride)
    assert _err(got, want) < F32_GRADE_TOL
    e32 = _err(ops.conv7x7(x, ops.conv7x7_filter(w), K, stride=stride), want)                # float32 instruction, direct sum
    assert _err(got, want) < 3.0 * e32 + 2e-7
    stats = torch.zeros((N, K, 2), device=DEV, dtype=torch.float64)
    raw = ops.conv7x7_split(x, A, K, stride=stride, stats=stats)
    want_stats = torch.stack([want.sum((2, 3)), (want * want).sum((2, 3))], dim=2)
    assert (stats - want_stats).abs().max().item() < 1e-5 * max(1.0, want_stats.abs().max().item())
    scale, shift = _rand(K, seed=132).to(DEV), _rand(K, seed=133).to(DEV)
    ref = torch.relu(want * scale.double()[None, :, None, None] + shift.double()[None, :, None, None])
    assert _err(ops.conv7x7_split(x, A, K, scale, shift, True, stride), ref) < 2 * F32_GRADE_TOL


@pytest.mark.parametrize("N,C,H,W,K", [(2, 336, 60, 80, 256), (2, 128, 33, 70, 100), (1, 256, 15, 20, 576), (2, 64, 9, 132, 144)])
def test_conv1x1_float16_pieces(vo, N, C, H, W, K):
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=170).to(DEV)
    w = (_rand(K, C, 1, 1, seed=171) / C ** 0.5).to(DEV)
    bias = _rand(K, seed=172).to(DEV)
    A = ops.conv1x1_split_filter(w, "f16")
    want = Fn.conv2d(x.double(), w.double(), None)
    got = ops.conv1x1_split(x, A, K)
    assert _err(got, want) < F32_GRADE_TOL
    # the kernel with the weights resident in LDS (where the layer fits it) and the template's kernel: the same bits
    assert torch.equal(got, ops.conv1x1_split(x, A, K, use_template=True))
    sc = (_rand(K, seed=173).abs() + 0.5).to(DEV)
    assert torch.equal(ops.conv1x1_split(x, A, K, sc, bias, True), ops.conv1x1_split(x, A, K, sc, bias, True, use_template=True))
    e32 = _err(ops.conv1x1(x, ops.conv1x1_filter(w), K), want)                               # float32 instruction, direct sum
    assert _err(got, want) < 3.0 * e32 + 2e-7
    pm = ops.conv1x1_pixel_major_split(x, A, K, None, bias, False)
    assert _err(pm.permute(0, 3, 1, 2), want + bias.double()[None, :, None, None]) < 2 * F32_GRADE_TOL
    with pytest.raises(ValueError):
        ops.conv1x1_pixel_major_split(x, A, K, None, None, False, split_rows=True)      # rows split beforehand are bf16 pieces


@pytest.mark.parametrize("ks", [(3, 3), (1, 5), (5, 1)])
@pytest.mark.parametrize("B,C,X,H,W", [(2, 128, 256, 60, 80), (1, 64, 192, 24, 64)])
def test_gru_step_float16_pieces(vo, B, C, X, H, W, ks):
    from visual_odometry_amd import ops
    pad = (ks[0] // 2, ks[1] // 2)
    h = torch.tanh(_rand(B, C, H, W, seed=200)).to(DEV)
    x = _rand(B, X, H, W, seed=201).to(DEV)
    wz, wr, wq = ((_rand(C, C + X, *ks, seed=202 + i) / (ks[0] * ks[1] * (C + X)) ** 0.5).to(DEV) for i in range(3))
    bz, br, bq = (_rand(C, seed=205 + i).to(DEV) for i in range(3))
    ctx = tuple((_rand(B, 3 * C, H, W, seed=208) * 0.5).to(DEV).split(C, dim=1))
    buf = torch.cat([h, x], dim=1).contiguous()
    hx = buf.double()
    z = torch.sigmoid(Fn.conv2d(hx, wz.double(), bz.double(), 1, pad) + ctx[0].double())
    r = torch.sigmoid(Fn.conv2d(hx, wr.double(), br.double(), 1, pad) + ctx[1].double())
    q = torch.tanh(Fn.conv2d(torch.cat([r * h.double(), x.double()], dim=1), wq.double(), bq.double(), 1, pad) + ctx[2].double())
    want = (1 - z) * h.double() + z * q
    Uzr = ops.gru_split_filter(torch.cat([wz, wr], dim=0).contiguous(), "f16")
    Uq = ops.gru_split_filter(wq, "f16")
    zg, rh = ops.gru_conv_zr(buf, Uzr, torch.cat([bz, br]).contiguous(), ctx[0], ctx[1], h, ks)
    assert (zg.double() - z).abs().max().item() < 2e-6 and (rh.double() - r * h.double()).abs().max().item() < 2e-6
    got = ops.gru_conv_q(buf, rh, Uq, bq, ctx[2], zg, h, ks, h_copy=buf[:, :C])
    assert (got.double() - want).abs().max().item() < 4e-6
    assert torch.equal(buf[:, :C], got) and torch.equal(buf[:, C:], x)


def test_two_pipelines_in_one_process_keep_their_own_arithmetic(vo):
    """The arithmetic is a field of each model, put in force only inside that model's forward pass: a bf16x3
    pipeline and an f16x3 pipeline interleaved in one process give what each gives alone."""
    import numpy as np
    from visual_odometry_amd import synthetic as syn
    from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
    from visual_odometry_amd.raft.models import RAFTStereoConfig
    H, W = 128, 160
    make = lambda kind: VisualOdometry(VOConfig(iters=3, stereo=RAFTStereoConfig.realtime(), conv_arithmetic=kind), syn.left_K(W / 640.0),
                                       syn.focal_length_px() * W / 640.0, syn.baseline(), np.ones((H, W), bool), seed=0)
    left, right = syn.stereo_sequence(seed=3, frames=3, h=H, w=W)
    left, right = torch.from_numpy(left).to(DEV), torch.from_numpy(right).to(DEV)
    a, b = make("bf16x3"), make("f16x3")
    fa0, fb0 = a.flows(left), b.flows(left)
    fa1 = a.flows(left); fb1 = b.flows(left); fa2 = a.flows(left)
    assert torch.equal(fa0, fa1) and torch.equal(fa0, fa2) and torch.equal(fb0, fb1)
    assert not torch.equal(fa0, fb0)
    with pytest.raises(ValueError):
        make("int8")


# ---------------------------------------------------------------------------------------------------
# correlation volumes on float16 pieces (per-image power-of-two scaling from the largest |x|)

def _amax_bits_host(x: torch.Tensor) -> torch.Tensor:
    """Host statement of vo_feature_rows_amax: the float bits of each image's largest |x|."""
    return x.abs().reshape(x.shape[0], -1).amax(dim=1).contiguous().view(torch.int32)


def _feature_exp_host(bits: torch.Tensor) -> torch.Tensor:
    E = (bits >> 23) & 0xff
    e = torch.clamp(140 - E, max=120)
    return torch.where((E == 0) | (E == 255), torch.zeros_like(e), e)


def _split_rows_f16_host(x: torch.Tensor, block: int, bits: torch.Tensor) -> torch.Tensor:
    """Host statement of the float16 split-row format: x' = x * 2^e(image), high = float16(x'), low =
    float16(x' - high), per row and block ``block`` high parts then ``block`` low parts; int16 bits (..., 2*D)."""
    e = _feature_exp_host(bits).view(-1, *([1] * (x.dim() - 1)))
    xs = x * torch.exp2(e.float())
    hi = xs.to(torch.float16)
    lo = (xs - hi.float()).to(torch.float16)
    D = x.shape[-1]
    hb = hi.view(torch.int16).reshape(*x.shape[:-1], D // block, block)
    lb = lo.view(torch.int16).reshape(*x.shape[:-1], D // block, block)
    return torch.cat([hb, lb], dim=-1).reshape(*x.shape[:-1], 2 * D)


@pytest.mark.parametrize("B,H,W,D", [(2, 12, 20, 256), (1, 7, 10, 64), (1, 9, 65, 256), (2, 60, 80, 256), (3, 33, 40, 128)])
def test_corr_volume_2d_float16_pieces(vo, B, H, W, D):
    """The flow volume on float16 pieces: the per-image maxima and the cut rows against their host statements (bit for
    bit), the kernel that reads cut rows against the kernel that cuts inside (bit for bit), both against the float64
    volume at the float32 instruction's own error, and no worse than twice the float32-instruction kernel."""
    from visual_odometry_amd import ops
    f1, f2 = _rand(B, H, W, D, seed=31).to(DEV), (_rand(B, H, W, D, seed=32) * 3.7).to(DEV)
    a1, a2 = ops.feature_rows_amax(f1), ops.feature_rows_amax(f2)
    assert torch.equal(a1.cpu(), _amax_bits_host(f1.cpu())) and torch.equal(a2.cpu(), _amax_bits_host(f2.cpu()))
    s1, s2 = ops.split_feature_rows(f1, 32, "f16", a1), ops.split_feature_rows(f2, 32, "f16", a2)
    assert torch.equal(s1.view(torch.int16).cpu(), _split_rows_f16_host(f1.cpu(), 32, a1.cpu()))
    got = ops.corr_volume_2d_presplit(s1, s2, "f16", a1, a2)
    assert torch.equal(got, ops.corr_volume_2d(f1, f2, split="f16"))
    assert torch.equal(got, ops.corr_volume_2d(f1, f2, split="f16", amax1=a1, amax2=a2))
    want = torch.einsum("bid,bjd->bij", f1.view(B, H * W, D).double(), f2.view(B, H * W, D).double()) / D ** 0.5
    e16 = _err(got.view(B, H * W, H * W), want)
    e32 = _err(ops.corr_volume_2d(f1, f2).view(B, H * W, H * W), want)
    ebf = _err(ops.corr_volume_2d(f1, f2, split=True).view(B, H * W, H * W), want)
    assert e16 < F32_GRADE_TOL and e16 < 2 * e32 + 2e-7, (e16, e32)
    assert ebf > BF16_FLOOR > e16, (ebf, e16)


def test_corr_volume_float16_pieces_scale_with_powers_of_two_and_per_image(vo):
    """Multiplying a map by a power of two multiplies the volume by it bit for bit, from 2^-60 to 2^60 (the per-image
    exponent absorbs it); a pair's volume does not depend on the other pairs of its launch; maps whose images differ in
    scale by 10^6 keep the float32-grade error per image; small integers are exact; an all-zero map gives zeros and a
    NaN in a map stays a NaN in its own image only."""
    from visual_odometry_amd import ops
    B, H, W, D = 3, 20, 24, 256
    f1, f2 = _rand(B, H, W, D, seed=33).to(DEV), _rand(B, H, W, D, seed=34).to(DEV)
    vol = ops.corr_volume_2d(f1, f2, split="f16")
    for k1, k2 in ((7, -3), (-60, 60), (40, 20), (-50, -50)):
        assert torch.equal(ops.corr_volume_2d(f1 * 2.0 ** k1, f2 * 2.0 ** k2, split="f16"), vol * 2.0 ** (k1 + k2)), (k1, k2)
    assert torch.equal(ops.corr_volume_2d(f1[1:2].contiguous(), f2[1:2].contiguous(), split="f16"), vol[1:2])
    scales = torch.tensor([1e-3, 1.0, 1e3], device=DEV).view(3, 1, 1, 1)
    g = ops.corr_volume_2d(f1 * scales, f2, split="f16").view(B, H * W, H * W)
    want = torch.einsum("bid,bjd->bij", (f1 * scales).view(B, H * W, D).double(), f2.view(B, H * W, D).double()) / 16.0
    for b in range(B):
        assert _err(g[b], want[b]) < F32_GRADE_TOL
    gi = torch.Generator().manual_seed(35)
    i1 = torch.randint(-3, 4, (1, 9, 65, 256), generator=gi).float()
    i2 = torch.randint(-3, 4, (1, 9, 65, 256), generator=gi).float()
    exact = (i1.view(585, 256).double() @ i2.view(585, 256).double().T / 16.0).float()
    assert torch.equal(ops.corr_volume_2d(i1.to(DEV), i2.to(DEV), split="f16").cpu().view(585, 585), exact)
    z = f1.clone(); z[0] = 0.0; z[2, 3, 4, 5] = float("nan")
    vz = ops.corr_volume_2d(z, f2, split="f16")
    assert torch.equal(vz[0], torch.zeros_like(vz[0])) and torch.equal(vz[1], vol[1])
    assert torch.isnan(vz[2].view(H * W, H * W)[3 * W + 4]).all()


@pytest.mark.parametrize("B,H,W1,W2,D", [(2, 6, 160, 160, 256), (1, 3, 40, 40, 256), (1, 2, 320, 320, 256), (2, 5, 50, 37, 64)])
def test_corr_volume_1d_float16_pieces(vo, B, H, W1, W2, D):
    """The per-row (stereo) volume on float16 pieces, one exponent per image shared by its rows: against float64, against
    the kernel on rows cut beforehand (bit for bit), exact power-of-two scaling."""
    from visual_odometry_amd import ops
    f1, f2 = (_rand(B, H, W1, D, seed=36) * 0.05).to(DEV), _rand(B, H, W2, D, seed=37).to(DEV)
    got = ops.corr_volume_1d(f1, f2, split="f16")
    want = torch.einsum("bhid,bhjd->bhij", f1.double(), f2.double()) / D ** 0.5
    e16, e32 = _err(got, want), _err(ops.corr_volume_1d(f1, f2), want)
    assert got.shape == (B, H, W1, W2) and e16 < F32_GRADE_TOL and e16 < 2 * e32 + 2e-7, (e16, e32)
    a1, a2 = ops.feature_rows_amax(f1), ops.feature_rows_amax(f2)
    pre = ops.corr_volume_1d_presplit(ops.split_feature_rows(f1, 16, "f16", a1), ops.split_feature_rows(f2, 16, "f16", a2), "f16", a1, a2)
    assert torch.equal(pre, got)
    assert torch.equal(ops.corr_volume_1d(f1 * 1024.0, f2 * 2.0 ** -9, split="f16"), got * 2.0)
    with pytest.raises(ValueError):
        ops.corr_volume_1d_presplit(f1, f2, "f16")                       # cut rows without their maxima
    with pytest.raises(ValueError):
        ops.corr_volume_1d(f1, f2, split="fp8")


def test_flow_volume_of_consecutive_pairs_cuts_each_frame_once(vo):
    """raft.models._flow_volume_f16 on the maps of a sequence (maps[:-1], maps[1:], views of one buffer) cuts the B+1
    frames once; the volume equals the one from two separate tensors bit for bit."""
    from visual_odometry_amd import ops
    from visual_odometry_amd.raft import models
    maps = _rand(5, 12, 20, 256, seed=38).to(DEV)
    rec = {}
    old = ops.set_recorder(rec)
    try:
        got = models._flow_volume_f16(maps[:-1], maps[1:])
    finally:
        ops.set_recorder(old)
    assert len(rec["vo_feature_rows_amax"]) == 1 and len(rec["vo_split_feature_rows"]) == 1
    a, b = maps[:-1].clone(), maps[1:].clone()
    assert torch.equal(got, models._flow_volume_f16(a, b))
    assert torch.equal(got, ops.corr_volume_2d(a, b, split="f16"))
    # a view that ends at the buffer's end must not be extended past it
    assert torch.equal(models._flow_volume_f16(maps[:4], maps[:4]), ops.corr_volume_2d(maps[:4].contiguous(), maps[:4].contiguous(), split="f16"))


def test_corr_volume_float16_pieces_over_odd_shapes(vo):
    """Seeded random shapes that leave the fast paths: pixel counts that are no multiple of 4 (element-wise stores), feature
    widths that are multiples of 32 but not of 64 (the small tile with the large volume), single rows and columns,
    different widths of the two stereo maps -- every one against float64 and, where rows can be cut beforehand, bit for
    bit against that path."""
    import random
    from visual_odometry_amd import ops
    rnd = random.Random(7)
    for case in range(10):
        B, H, W, D = rnd.randint(1, 3), rnd.randint(1, 23), rnd.randint(1, 41), 32 * rnd.randint(1, 8)
        f1 = (_rand(B, H, W, D, seed=50 + case) * 10.0 ** rnd.uniform(-3, 3)).to(DEV)
        f2 = (_rand(B, H, W, D, seed=70 + case) * 10.0 ** rnd.uniform(-3, 3)).to(DEV)
        got = ops.corr_volume_2d(f1, f2, split="f16")
        want = torch.einsum("bid,bjd->bij", f1.view(B, H * W, D).double(), f2.view(B, H * W, D).double()) / D ** 0.5
        assert _err(got.view(B, H * W, H * W), want) < F32_GRADE_TOL, (B, H, W, D)
        if D % 64 == 0:
            a1, a2 = ops.feature_rows_amax(f1), ops.feature_rows_amax(f2)
            pre = ops.corr_volume_2d_presplit(ops.split_feature_rows(f1, 32, "f16", a1), ops.split_feature_rows(f2, 32, "f16", a2), "f16", a1, a2)
            assert torch.equal(pre, got), (B, H, W, D)
        W2 = rnd.randint(1, 41)
        g2 = (_rand(B, H, W2, D, seed=90 + case) * 10.0 ** rnd.uniform(-2, 2)).to(DEV)
        got1 = ops.corr_volume_1d(f1, g2, split="f16")
        want1 = torch.einsum("bhid,bhjd->bhij", f1.double(), g2.double()) / D ** 0.5
        assert got1.shape == (B, H, W, W2) and _err(got1, want1) < F32_GRADE_TOL, (B, H, W, W2, D)
