"""HIP correlation / pyramid / lookup / upsample kernels (through the C ABI)
against the stock-PyTorch oracle, operator by operator and end to end.

Parity for this half is UNPINNED by the reference (its RAFT submodules are
empty; see oracle/raft_ref.py): the oracle is this repository's restatement.
Tolerances: volume / lookup / upsample <= 1e-4 absolute on O(1) values,
end-to-end flow / disparity <= 0.05 px mean end-point error (north star).
"""
import numpy as np
import pytest
import torch

from oracle import raft_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _nchw(pm):          # pixel-major (B,H,W,D) -> (B,D,H,W)
    return pm.permute(0, 3, 1, 2).contiguous()


# ---------------------------------------------------------------- correlation

@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("B,H,W,D", [(2, 12, 20, 256), (1, 7, 10, 64), (1, 16, 16, 32), (1, 60, 80, 256)])
def test_corr_volume_2d(vo, B, H, W, D, split):
    """Both arithmetics (float32 matrix instruction; three bf16 terms per product) against the float64
    restatement: 2e-5 absolute on entries of order one (split measured 2-5e-6)."""
    from visual_odometry_amd import ops
    f1, f2 = _rand(B, H, W, D, seed=1), _rand(B, H, W, D, seed=2)
    got = ops.corr_volume_2d(f1.to(DEV), f2.to(DEV), split=split)
    assert got.shape == (B, H * W, H, W)
    want = raft_ref.corr_volume_2d(_nchw(f1).double(), _nchw(f2).double()).float()
    assert (got.cpu() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
    if not split:
        # transposing the roles transposes the volume bit for bit (same k order, a*b == b*a)
        swapped = ops.corr_volume_2d(f2.to(DEV), f1.to(DEV))
        assert torch.equal(swapped.view(B, H * W, H * W).transpose(1, 2), got.view(B, H * W, H * W))
    else:
        # features spanning many binades: every product keeps 16 significant bits
        s1 = (f1 * torch.exp2(torch.randint(-10, 10, f1.shape).float())).to(DEV)
        gots = ops.corr_volume_2d(s1, f2.to(DEV), split=True)
        wants = raft_ref.corr_volume_2d(_nchw(s1.cpu()).double(), _nchw(f2).double())
        assert (gots.cpu().double() - wants).abs().max().item() < 2e-5 * max(1.0, wants.abs().max().item())
        assert torch.equal(gots, ops.corr_volume_2d(s1, f2.to(DEV), split=True))


@pytest.mark.parametrize("split", [False, True])
def test_corr_volume_exact_on_integers(vo, split):
    """Small-integer features make every product and partial sum exact in float32 (and their high bf16
    parts hold them whole, so the split arithmetic is exact too):
    any operand-layout or tile-index slip shows up as a wrong integer.  fmap2 is
    not symmetric in anything, so a transposed write cannot hide."""
    from visual_odometry_amd import ops
    B, H, W, D = 1, 9, 65, 256            # 585 pixels: five 128-tiles, the last partial
    g = torch.Generator().manual_seed(3)
    f1 = torch.randint(-3, 4, (B, H, W, D), generator=g).float()
    f2 = torch.randint(-3, 4, (B, H, W, D), generator=g).float()
    got = ops.corr_volume_2d(f1.to(DEV), f2.to(DEV), split=split).cpu().view(H * W, H * W)
    want = (f1.view(H * W, D).double() @ f2.view(H * W, D).double().T / 16.0).float()
    assert torch.equal(got, want)
    # 1-D flavour, widths that are not multiples of the 32-wide MFMA tile
    f1 = torch.randint(-3, 4, (2, 5, 50, 64), generator=g).float()
    f2 = torch.randint(-3, 4, (2, 5, 37, 64), generator=g).float()
    got = ops.corr_volume_1d(f1.to(DEV), f2.to(DEV), split=split).cpu()
    want = (torch.einsum("bhid,bhjd->bhij", f1.double(), f2.double()) / 8.0).float()
    assert got.shape == (2, 5, 50, 37) and torch.equal(got, want)


@pytest.mark.parametrize("B,H,W,D", [(2, 6, 160, 256), (1, 3, 40, 256), (1, 2, 320, 256), (1, 2, 200, 128)])
@pytest.mark.parametrize("split", [False, True])
def test_corr_volume_1d(vo, B, H, W, D, split):
    from visual_odometry_amd import ops
    f1, f2 = _rand(B, H, W, D, seed=4), _rand(B, H, W, D, seed=5)
    got = ops.corr_volume_1d(f1.to(DEV), f2.to(DEV), split=split)
    want = raft_ref.corr_volume_1d(_nchw(f1).double(), _nchw(f2).double()).float()
    assert got.shape == (B, H, W, W)
    assert (got.cpu() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


def _split_rows_host(x: torch.Tensor, block: int) -> torch.Tensor:
    """Host statement of the split-row format: per row and block of ``block`` features, ``block`` high parts
    (bfloat16(x), round to nearest even) then ``block`` low parts (bfloat16(x - high)); returned as int16 bits
    (..., 2*D)."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    D = x.shape[-1]
    hb = hi.view(torch.int16).reshape(*x.shape[:-1], D // block, block)
    lb = lo.view(torch.int16).reshape(*x.shape[:-1], D // block, block)
    return torch.cat([hb, lb], dim=-1).reshape(*x.shape[:-1], 2 * D)


@pytest.mark.parametrize("B,H,W,D", [(2, 12, 20, 256), (1, 7, 10, 64), (1, 9, 65, 256), (1, 60, 80, 256), (3, 33, 40, 128)])
def test_corr_volume_2d_on_rows_split_beforehand(vo, B, H, W, D):
    """vo_split_feature_rows against the host statement of its format (bit for bit), and the volume kernel
    for such rows (256 x 256 tiles, persistent workgroups; maps smaller and larger than a tile, partial
    tiles, several tiles per workgroup) against the kernel that splits inside (bit for bit: the same
    products in the same order) and against the float64 restatement."""
    from visual_odometry_amd import ops
    f1, f2 = _rand(B, H, W, D, seed=21).to(DEV), _rand(B, H, W, D, seed=22).to(DEV)
    s1, s2 = ops.split_feature_rows(f1, 32), ops.split_feature_rows(f2, 32)
    assert torch.equal(s1.view(torch.int16).cpu(), _split_rows_host(f1.cpu(), 32))
    got = ops.corr_volume_2d_presplit(s1, s2)
    assert torch.equal(got, ops.corr_volume_2d(f1, f2, split=True))
    want = raft_ref.corr_volume_2d(_nchw(f1.cpu()).double(), _nchw(f2.cpu()).double()).float()
    assert (got.cpu() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
    assert torch.equal(got, ops.corr_volume_2d_presplit(s1, s2))


def test_corr_volume_presplit_exact_on_integers_and_1d(vo):
    """Small integers make every product exact (see test_corr_volume_exact_on_integers): a slip in the row
    format, the tile list or the store path shows as a wrong integer.  Sizes: 585 and 1300 pixels (three and
    six 256-tiles per side, the last partial), and the 1-D flavour on blocks of 16."""
    from visual_odometry_amd import ops
    g = torch.Generator().manual_seed(23)
    for (H, W) in ((9, 65), (26, 50)):
        f1 = torch.randint(-3, 4, (2, H, W, 256), generator=g).float()
        f2 = torch.randint(-3, 4, (2, H, W, 256), generator=g).float()
        got = ops.corr_volume_2d_presplit(ops.split_feature_rows(f1.to(DEV), 32), ops.split_feature_rows(f2.to(DEV), 32))
        want = (torch.einsum("bid,bjd->bij", f1.view(2, H * W, 256).double(), f2.view(2, H * W, 256).double()) / 16.0).float()
        assert torch.equal(got.cpu().view(2, H * W, H * W), want)
    f1 = torch.randint(-3, 4, (2, 5, 50, 64), generator=g).float()
    f2 = torch.randint(-3, 4, (2, 5, 37, 64), generator=g).float()
    r1, r2 = ops.split_feature_rows(f1.to(DEV), 16), ops.split_feature_rows(f2.to(DEV), 16)
    assert torch.equal(r1.view(torch.int16).cpu(), _split_rows_host(f1, 16))
    got = ops.corr_volume_1d_presplit(r1, r2).cpu()
    want = (torch.einsum("bhid,bhjd->bhij", f1.double(), f2.double()) / 8.0).float()
    assert got.shape == (2, 5, 50, 37) and torch.equal(got, want)
    h1, h2 = _rand(2, 6, 160, 256, seed=24).to(DEV), _rand(2, 6, 160, 256, seed=25).to(DEV)
    assert torch.equal(ops.corr_volume_1d_presplit(ops.split_feature_rows(h1, 16), ops.split_feature_rows(h2, 16)),
                       ops.corr_volume_1d(h1, h2, split=True))
    with pytest.raises(ValueError):
        ops.split_feature_rows(torch.zeros(4, 48, device=DEV), 32)              # D not a multiple of the block


def test_lookup_refuses_sizes_beyond_its_32_bit_addressing(vo):
    """The lookups address a workgroup's 256 slices and one pair's output with 32-bit offsets; a target map whose 256
    slices exceed 2 GiB is refused before anything is launched (the pointers here are never dereferenced)."""
    from visual_odometry_amd import _lib
    t = torch.zeros(64, device=DEV)
    c = torch.zeros(2 * 4, device=DEV)
    with pytest.raises(ValueError, match="2 GiB"):
        _lib.call("vo_corr_lookup_2d", t.data_ptr(), None, None, None, c.data_ptr(), t.data_ptr(), 0, 1, 2, 2, 1500, 1500, 1, 4, None)
    with pytest.raises(ValueError, match="2 GiB"):
        _lib.call("vo_corr_lookup_1d", t.data_ptr(), None, None, None, c.data_ptr(), t.data_ptr(), 0, 1, 2, 2, 3000000, 1, 4, None)


def test_correlation_chain_properties_at_full_size(vo):
    """The 640x480 geometry (60x80 maps of 256 features, 16 pairs, 1.5 GB of volume), where no CPU restatement
    finishes in seconds: properties that need none.  Scaling a map by a power of two scales the volume exactly
    (bit for bit, both kernels); swapping the maps transposes it (to the rounding of the two small terms, which
    swap their order); every pyramid level has the mean of the one below where its size is even; a lookup at
    integer coordinates returns volume entries as they are, zeros outside."""
    from visual_odometry_amd import ops
    B, H, W, D = 16, 60, 80, 256
    N = H * W
    f1, f2 = _rand(B, H, W, D, seed=301).to(DEV), _rand(B, H, W, D, seed=302).to(DEV)
    s1, s2 = ops.split_feature_rows(f1, 32), ops.split_feature_rows(f2, 32)
    vol = ops.corr_volume_2d_presplit(s1, s2)
    assert torch.equal(vol, ops.corr_volume_2d(f1, f2, split=True))
    assert torch.equal(ops.corr_volume_2d_presplit(ops.split_feature_rows(f1 * 4.0, 32), s2), vol * 4.0)
    # the float16-piece flavour (the default arithmetic's): rows cut beforehand = cut inside, power-of-two scaling exact,
    # within float32 rounding of the bf16-piece volume's neighbourhood (5e-6) and of the float32-instruction volume (1e-6)
    a1, a2 = ops.feature_rows_amax(f1), ops.feature_rows_amax(f2)
    v16 = ops.corr_volume_2d_presplit(ops.split_feature_rows(f1, 32, "f16", a1), ops.split_feature_rows(f2, 32, "f16", a2), "f16", a1, a2)
    assert torch.equal(v16, ops.corr_volume_2d(f1, f2, split="f16"))
    assert torch.equal(ops.corr_volume_2d(f1 * 0.125, f2 * 64.0, split="f16"), v16 * 8.0)
    top = vol.abs().max().item()
    assert (v16 - ops.corr_volume_2d(f1, f2)).abs().max().item() < 1.5e-6 * top and (v16 - vol).abs().max().item() < 1e-5 * top
    del v16
    swapped = ops.corr_volume_2d_presplit(s2, s1).view(B, N, N).transpose(1, 2)
    assert (swapped - vol.view(B, N, N)).abs().max().item() < 2e-6 * top
    del swapped
    pyr = ops.corr_pyramid_2d(vol, 4)
    assert [tuple(p.shape[2:]) for p in pyr] == [(60, 80), (30, 40), (15, 20), (7, 10)]
    m = [p.double().mean(dim=(2, 3)) for p in pyr[:3]]
    assert (m[1] - m[0]).abs().max().item() < 1e-6 * top and (m[2] - m[1]).abs().max().item() < 1e-6 * top
    # level 3 drops the odd last row of level 2: its mean is the mean of the 14 rows it covers
    m2 = pyr[2][:, :, :14].double().mean(dim=(2, 3))
    assert (pyr[3].double().mean(dim=(2, 3)) - m2).abs().max().item() < 1e-6 * top
    coords = raft_ref.coords_grid(B, H, W).to(DEV)
    win = ops.corr_lookup_2d(pyr[:1], coords, 4)                      # (B, 81, H, W): channel = x-offset index * 9 + y-offset index
    v5 = vol.view(B, H, W, H, W)
    ys, xs = torch.meshgrid(torch.arange(H, device=DEV), torch.arange(W, device=DEV), indexing="ij")
    for (dx, dy) in ((0, 0), (-4, -4), (4, 4), (3, -2), (-1, 4)):
        ty, tx = ys + dy, xs + dx
        ok = (ty >= 0) & (ty < H) & (tx >= 0) & (tx < W)
        want = torch.where(ok[None], v5[:, ys, xs, ty.clamp(0, H - 1), tx.clamp(0, W - 1)], torch.zeros((), device=DEV))
        assert torch.equal(win[:, (dx + 4) * 9 + (dy + 4)], want)


def test_corr_rejects_bad_input(vo):
    from visual_odometry_amd import ops
    from visual_odometry_amd._lib import VoHipError
    with pytest.raises(VoHipError):          # feature dim not a multiple of 32
        ops.corr_volume_2d(torch.zeros(1, 4, 4, 48, device=DEV), torch.zeros(1, 4, 4, 48, device=DEV))
    with pytest.raises(ValueError):
        ops.corr_volume_2d(torch.zeros(1, 4, 4, 64), torch.zeros(1, 4, 4, 64))      # CPU tensors
    with pytest.raises(ValueError):
        ops.corr_volume_2d(torch.zeros(1, 4, 4, 64, device=DEV), torch.zeros(1, 4, 5, 64, device=DEV))


# -------------------------------------------------------------------- pyramid

@pytest.mark.parametrize("B,N,H,W", [(1, 50, 60, 80), (2, 7, 15, 21), (1, 3, 9, 8), (1, 5, 90, 160), (1, 4, 6, 7)])
def test_pyramid_2d(vo, B, N, H, W):
    from visual_odometry_amd import ops
    levels = 4 if min(H, W) >= 8 else 3 if min(H, W) >= 4 else 2
    c = _rand(B, N, H, W, seed=6)
    got = ops.corr_pyramid_2d(c.to(DEV), levels)
    want = raft_ref.corr_pyramid_2d(c, levels)
    assert len(got) == levels
    for g, w in zip(got, want):
        assert g.shape == w.shape
        torch.testing.assert_close(g.cpu(), w, rtol=1e-6, atol=1e-6)
    # averaging is linear: pyramid(a + 2b) == pyramid(a) + 2 pyramid(b) up to rounding
    d = _rand(B, N, H, W, seed=7)
    pd = ops.corr_pyramid_2d(d.to(DEV), levels)
    ps = ops.corr_pyramid_2d((c + 2 * d).to(DEV), levels)
    for a, b, s in zip(got, pd, ps):
        torch.testing.assert_close(s, a + 2 * b, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,H,W1,W2", [(1, 4, 160, 160), (2, 3, 10, 37), (1, 2, 5, 320), (1, 1, 3, 9)])
def test_pyramid_1d(vo, B, H, W1, W2):
    from visual_odometry_amd import ops
    c = _rand(B, H, W1, W2, seed=8)
    got = ops.corr_pyramid_1d(c.to(DEV), 4)
    want = raft_ref.corr_pyramid_1d(c, 4)
    for g, w in zip(got, want):
        assert g.shape == w.shape
        torch.testing.assert_close(g.cpu(), w, rtol=1e-6, atol=1e-6)


# --------------------------------------------------------------------- lookup

def _coords(B, H, W, seed, spread):
    base = raft_ref.coords_grid(B, H, W)
    c = base + _rand(B, 2, H, W, seed=seed, scale=spread)
    # hand-placed cases: exact integers, just outside each border, far outside
    c[0, :, 0, 0] = torch.tensor([0.0, 0.0])
    c[0, :, 0, 1] = torch.tensor([-0.5, -0.5])
    c[0, :, 0, 2] = torch.tensor([W - 1.0, H - 1.0])
    c[0, :, 0, 3] = torch.tensor([W - 0.5, H + 3.25])
    c[0, :, 0, 4] = torch.tensor([-100.0, 50.0])
    c[0, :, 0, 5] = torch.tensor([3.0, 1e6])
    return c


@pytest.mark.parametrize("B,H,W,radius,levels", [(1, 16, 24, 4, 4), (2, 8, 16, 3, 3), (1, 60, 80, 4, 4), (2, 9, 13, 4, 3), (1, 7, 11, 3, 2)])
def test_lookup_2d(vo, B, H, W, radius, levels):
    """Sizes with and without a multiple of four pixels, and the output written as the leading channels of a
    wider tensor (a batch stride)."""
    from visual_odometry_amd import ops
    N = H * W
    if N > 1000:                      # full 640x480 geometry: build the volume on the GPU
        f1, f2 = _rand(B, H, W, 64, seed=9), _rand(B, H, W, 64, seed=10)
        pyr = ops.corr_pyramid_2d(ops.corr_volume_2d(f1.to(DEV), f2.to(DEV)), levels)
    else:
        pyr = ops.corr_pyramid_2d(_rand(B, N, H, W, seed=9).to(DEV), levels)
    coords = _coords(B, H, W, 11, 3.0)
    got = ops.corr_lookup_2d(pyr, coords.to(DEV), radius)
    assert got.shape == (B, levels * (2 * radius + 1) ** 2, H, W)
    # float64 oracle: removes the normalise/un-normalise rounding of grid_sample
    want = raft_ref.corr_lookup_2d([p.cpu().double() for p in pyr], coords.double(), radius)
    assert (got.cpu().double() - want).abs().max().item() < 2e-5
    # and the float32 stock operator itself agrees to its own rounding
    want32 = raft_ref.corr_lookup_2d([p.cpu() for p in pyr], coords, radius)
    assert (got.cpu() - want32).abs().max().item() < 5e-4
    # integer coordinates reproduce volume entries exactly (weights 1 and 0)
    ci = raft_ref.coords_grid(B, H, W)
    exact = ops.corr_lookup_2d(pyr[:1], ci.to(DEV), radius).cpu()
    centre = radius * (2 * radius + 1) + radius
    diag = pyr[0].cpu().view(B, N, N).diagonal(dim1=1, dim2=2).reshape(B, H, W)
    assert torch.equal(exact[:, centre], diag)
    # written into the leading channels of a wider tensor: same values, the other channels untouched
    ch = got.shape[1]
    wide = torch.full((B, ch + 5, H, W), 7.5, device=DEV)
    ops.corr_lookup_2d(pyr, coords.to(DEV), radius, out=wide[:, :ch])
    assert torch.equal(wide[:, :ch], got) and torch.all(wide[:, ch:] == 7.5)


@pytest.mark.parametrize("B,H,W,radius", [(1, 6, 40, 4), (2, 4, 37, 3), (1, 120, 160, 4), (1, 5, 37, 4)])
def test_lookup_1d(vo, B, H, W, radius):
    from visual_odometry_amd import ops
    pyr = ops.corr_pyramid_1d(_rand(B, H, W, W, seed=12).to(DEV), 4)
    coords = _coords(B, H, W, 13, 5.0)
    got = ops.corr_lookup_1d(pyr, coords.to(DEV), radius)
    assert got.shape == (B, 4 * (2 * radius + 1), H, W)
    want = raft_ref.corr_lookup_1d([p.cpu().double() for p in pyr], coords.double(), radius)
    assert (got.cpu().double() - want).abs().max().item() < 2e-5


def test_lookup_rejects_mismatched_pyramid(vo):
    from visual_odometry_amd import ops
    pyr = ops.corr_pyramid_2d(torch.zeros(1, 48, 6, 8, device=DEV), 2)
    with pytest.raises(ValueError):
        ops.corr_lookup_2d(pyr, torch.zeros(1, 2, 6, 9, device=DEV))
    with pytest.raises(ValueError):
        ops.corr_lookup_2d([pyr[0], pyr[0]], torch.zeros(1, 2, 6, 8, device=DEV))
    from visual_odometry_amd._lib import VoHipError
    with pytest.raises(VoHipError):
        ops.corr_lookup_2d(pyr, torch.zeros(1, 2, 6, 8, device=DEV), radius=5)


# ------------------------------------------------------------------- upsample

@pytest.mark.parametrize("B,C,Cout,H,W,f", [(2, 2, 2, 12, 20, 8), (1, 2, 1, 30, 40, 4), (1, 2, 2, 5, 7, 2),
                                           (1, 2, 2, 60, 80, 8), (1, 2, 1, 120, 160, 4)])
def test_convex_upsample(vo, B, C, Cout, H, W, f):
    from visual_odometry_amd import ops
    flow = _rand(B, C, H, W, seed=14, scale=3.0)
    mask = _rand(B, 9 * f * f, H, W, seed=15, scale=4.0)
    got = ops.convex_upsample(flow.to(DEV), mask.to(DEV), f, out_channels=Cout, mask_scale=0.25)
    want = raft_ref.convex_upsample(flow, 0.25 * mask, f)[:, :Cout]
    assert got.shape == (B, Cout, f * H, f * W)
    assert (got.cpu() - want).abs().max().item() < 2e-5
    # a convex combination of a constant field is that constant (times f)
    const = torch.full((B, C, H, W), 1.5)
    up = ops.convex_upsample(const.to(DEV), mask.to(DEV), f, out_channels=Cout).cpu()
    inner = up[:, :, f:-f, f:-f]
    assert (inner - 1.5 * f).abs().max().item() < 1e-4


# ------------------------------------------------------------- whole networks

def _epe(a, b):
    return (a - b).pow(2).sum(1).sqrt().mean().item()


def _images(B, H, W, seed):
    from visual_odometry_amd import synthetic as syn
    left, right = syn.stereo_sequence(seed, B + 1, H, W)
    to = lambda a: torch.from_numpy(a).float()[:, None].expand(-1, 3, -1, -1).contiguous().to(DEV)
    return to(left), to(right)


@pytest.mark.parametrize("H,W,iters", [(128, 160, 4), (480, 640, 12)])
def test_raft_flow_end_to_end(vo, H, W, iters):
    from visual_odometry_amd.raft.models import RAFTFlow
    torch.manual_seed(0)
    model = RAFTFlow().to(DEV).eval()
    left, _ = _images(2, H, W, seed=5)
    lo, up = model(left[:2], left[1:3], iters=iters)
    lo_ref, up_ref = raft_ref.raft_flow_forward(model, left[:2], left[1:3], iters=iters)
    assert up.shape == (2, 2, H, W)
    assert _epe(up, up_ref) < 0.05 and _epe(lo, lo_ref) < 0.05 / 8
    assert float(up.abs().max()) > 1e-3        # the network actually moved
    # shared-frame feature maps (sequence mode) give the same flow
    from visual_odometry_amd.raft.models import normalize_image
    fm = model.features(normalize_image(left))
    lo2, up2 = model(left[:2], left[1:3], iters=iters, fmap1=fm[:2].contiguous(), fmap2=fm[1:3].contiguous())
    assert _epe(up2, up) < 1e-3


@pytest.mark.parametrize("H,W,iters,realtime", [(128, 160, 4, True), (128, 160, 3, False), (480, 640, 12, True),
                                                (480, 640, 12, False)])
def test_raft_stereo_end_to_end(vo, H, W, iters, realtime):
    """realtime False = the argument set the reference hard-codes (stereo/stereo_disparity_RAFT.py:33-44:
    hidden 128 x 3, three GRU levels); at 480x640 its coarsest maps are 15x20 and 30x40."""
    from visual_odometry_amd.raft.models import RAFTStereo, RAFTStereoConfig
    torch.manual_seed(1)
    model = RAFTStereo(RAFTStereoConfig.realtime() if realtime else RAFTStereoConfig()).to(DEV).eval()
    left, right = _images(1, H, W, seed=6)
    lo, disp = model(left, right, iters=iters)
    lo_ref, disp_ref = raft_ref.raft_stereo_forward(model, left, right, iters=iters)
    assert disp.shape == (2, 1, H, W)
    assert (disp - disp_ref).abs().mean().item() < 0.05
    assert float(lo[:, 1].abs().max()) == 0.0       # epipolar constraint: no y motion
    assert float(disp.abs().max()) > 1e-3


def test_networks_end_to_end_from_large_initial_displacements(vo):
    """Both networks started from a LARGE initial field (up to 20 cells = 160 px of flow, 30 cells = 120 px of
    disparity, many windows partly or wholly outside the volume, every pyramid level sampled far from its
    diagonal) against the restated networks started from the same field: the coordinates the lookups see are
    nothing like the near-identity ones of a zero start."""
    from visual_odometry_amd.raft.models import RAFTFlow, RAFTStereo, RAFTStereoConfig
    torch.manual_seed(21)
    flow, stereo = RAFTFlow().to(DEV).eval(), RAFTStereo(RAFTStereoConfig.realtime()).to(DEV).eval()
    H, W = 256, 320
    left, right = _images(2, H, W, seed=9)
    g = torch.Generator().manual_seed(5)
    f0 = ((torch.rand(2, 2, H // 8, W // 8, generator=g) - 0.5) * 40.0).to(DEV)
    f0[:, :, :2] += 25.0                                          # two rows of windows pushed past the border
    _, up = flow(left[:2], left[1:3], iters=5, flow_init=f0)
    lo_ref, up_ref = raft_ref.raft_flow_forward(flow, left[:2], left[1:3], iters=5, flow_init=f0)
    assert float(lo_ref.abs().max()) > 15.0                       # the field stayed large
    assert _epe(up, up_ref) < 1e-3 and (up - up_ref).abs().max().item() < 0.05
    d0 = torch.zeros(2, 2, H // 4, W // 4, device=DEV)
    d0[:, 0] = -((torch.rand(2, H // 4, W // 4, generator=g)) * 30.0).to(DEV)
    d0[:, 0, :, :5] -= 60.0                                       # columns whose windows lie left of the row
    _, disp = stereo(left[:2], right[:2], iters=5, flow_init=d0)
    lo_s, disp_ref = raft_ref.raft_stereo_forward(stereo, left[:2], right[:2], iters=5, flow_init=d0)
    assert float(lo_s[:, 0].abs().max()) > 40.0
    assert (disp - disp_ref).abs().mean().item() < 1e-3 and (disp - disp_ref).abs().max().item() < 0.05


def test_plugins_load_module_prefixed_checkpoints(vo, tmp_path):
    """The wrappers constructed from a checkpoint file saved the way the published models' are (every key prefixed
    ``module.``: flow/flow_map_RAFT.py:42-47, stereo/stereo_disparity_RAFT.py:46-51) compute with those weights:
    the same flow / disparity as the restated networks carrying the saved state dict."""
    from oracle import raft_modules
    from visual_odometry_amd import synthetic as syn
    from visual_odometry_amd.raft.models import RAFTFlow, RAFTStereo, RAFTStereoConfig
    from visual_odometry_amd.raft_plugins import HipDisparityRAFT, HipOpticalFlowRAFT
    torch.manual_seed(33)
    src_f, src_s = RAFTFlow().eval(), RAFTStereo(RAFTStereoConfig.realtime()).eval()
    for m in list(src_f.modules()) + list(src_s.modules()):
        if isinstance(m, torch.nn.BatchNorm2d):                    # a trained checkpoint's buffers are not the initial ones
            m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.6, 1.6)
    pf, ps = tmp_path / "raft_flow.pth", tmp_path / "raft_stereo.pth"
    torch.save({"module." + k: v for k, v in src_f.state_dict().items()}, pf)
    torch.save({"module." + k: v for k, v in src_s.state_dict().items()}, ps)
    left, right = syn.stereo_sequence(5, 2, 128, 160)
    fs = HipOpticalFlowRAFT(str(pf), None, iters=3, seed=99)       # the seed's random weights are overwritten by the file
    ds = HipDisparityRAFT(str(ps), None, iters=3, config=RAFTStereoConfig.realtime(), seed=98)
    for k, v in src_f.state_dict().items():
        assert torch.equal(fs.model.state_dict()[k].cpu(), v), k
    flow = fs.compute_flow(left[0], left[1])
    disp = ds.compute_disparity(left[0], right[0])
    rgb = lambda a: torch.from_numpy(np.repeat(a[None, None], 3, axis=1)).float().to(DEV)
    _, f_ref = raft_modules.flow_twin(src_f, DEV)(rgb(left[0]), rgb(left[1]), 3)
    _, d_ref = raft_modules.stereo_twin(src_s, DEV)(rgb(left[0]), rgb(right[0]), 3)
    assert np.abs(flow - f_ref[0].cpu().numpy()).max() < 1e-3 and np.abs(disp - d_ref[0, 0].cpu().numpy()).max() < 1e-3
    assert float(np.abs(flow).max()) > 1e-3


def test_gray_stem_equals_three_equal_channels(vo):
    """Encoders fed the single plane of a gray frame (stem filter summed over its input channels) against
    the same encoders fed three equal channels: rounding-level differences only."""
    from visual_odometry_amd.raft.layers import BasicEncoder, MultiBasicEncoder, fused_passes, gray_stem
    torch.manual_seed(3)
    for enc, kw in ((BasicEncoder(256, "instance", 3), {}), (BasicEncoder(256, "instance", 2), {}),
                    (MultiBasicEncoder([[64, 64, 64], [64, 64, 64]], "batch", 2), {"num_layers": 2})):
        enc = enc.to(DEV).eval()
        g = _rand(2, 1, 96, 128, seed=30).to(DEV)
        with torch.no_grad(), fused_passes():
            a = enc(g, **kw)
            b = enc(g.expand(-1, 3, -1, -1).contiguous(), **kw)
        flat = lambda o: torch.cat([t.flatten() for t in (o if not torch.is_tensor(o) else [o]) for t in (t if not torch.is_tensor(t) else [t])]) if not torch.is_tensor(o) else o.flatten()
        fa, fb = flat(a), flat(b)
        assert (fa - fb).abs().max().item() < 1e-4 * max(1.0, fb.abs().max().item())
        assert gray_stem(enc.conv1, g) is not enc.conv1 and gray_stem(enc.conv1, g) is gray_stem(enc.conv1, g)
        assert gray_stem(enc.conv1, g.expand(-1, 3, -1, -1)) is enc.conv1


def test_plugin_wrappers(vo):
    from visual_odometry_amd import synthetic as syn
    from visual_odometry_amd.raft.models import RAFTStereoConfig
    from visual_odometry_amd.raft_plugins import HipDisparityRAFT, HipOpticalFlowRAFT
    left, right = syn.stereo_sequence(3, 2, 100, 130)          # needs padding: 100x130 -> 104x136 / 128x160
    flow_solver = HipOpticalFlowRAFT(None, None, iters=2)
    flow = flow_solver.compute_flow(left[0], left[1])
    assert flow.shape == (2, 100, 130) and flow.dtype == np.float32
    bgr = np.repeat(left[0][:, :, None], 3, axis=2)
    assert np.allclose(flow_solver.compute_flow(bgr, np.repeat(left[1][:, :, None], 3, axis=2)), flow, atol=1e-4)
    with pytest.raises(ValueError):
        flow_solver.compute_flow(None, left[1])
    with pytest.raises(TypeError):
        flow_solver.compute_flow([1, 2], left[1])
    disp_solver = HipDisparityRAFT(None, None, iters=2, config=RAFTStereoConfig.realtime())
    disp = disp_solver.compute_disparity(left[0], right[0])
    assert disp.shape == (100, 130) and disp.dtype == np.float32
    with pytest.raises(ValueError):
        disp_solver.compute_disparity(left[0], None)
    with pytest.raises(TypeError):
        disp_solver.compute_disparity(left[0], 7)


@pytest.mark.parametrize("F,H,W,div", [(3, 480, 640, 8), (2, 100, 130, 8), (2, 100, 130, 32), (1, 720, 1280, 32), (2, 37, 52, 8)])
def test_prepare_images_matches_host_input_stage(vo, F, H, W, div):
    """uint8 gray -> 3 channels, replicate pad, 2*(x/255)-1: bit-exact with the
    wrapper's conversion + InputPadder + the models' entry normalisation."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    g = torch.Generator().manual_seed(F * H + W)
    gray = torch.randint(0, 256, (F, H, W), generator=g, dtype=torch.uint8)
    got, pads = ops.prepare_images(gray.to(DEV), div, normalize=True)
    rgb = gray.float()[:, None].expand(-1, 3, -1, -1).contiguous()
    ref_img, ref_pads = raft_ref.pad_to_multiple(rgb, div)
    assert pads == ref_pads and got.shape == ref_img.shape and got.shape[-1] % div == 0 and got.shape[-2] % div == 0
    assert torch.equal(got.cpu(), 2 * (ref_img / 255.0) - 1.0)
    raw, _ = ops.prepare_images(gray.to(DEV), div, normalize=False)
    assert torch.equal(raw.cpu(), ref_img)
    one, pads1 = ops.prepare_images(gray.to(DEV), div, normalize=True, channels=1)      # the plane the three channels equal
    assert pads1 == pads and one.shape[1] == 1 and torch.equal(one, got[:, :1])
    assert torch.equal(raft_ref.unpad(raw.cpu(), pads), rgb)


def test_sequence_entry_points_match_image_entry_points(vo):
    from visual_odometry_amd import synthetic as syn
    from visual_odometry_amd.raft.models import RAFTStereoConfig
    from visual_odometry_amd.raft_plugins import HipDisparityRAFT, HipOpticalFlowRAFT, gray_batch_to_rgb
    left, right = syn.stereo_sequence(4, 3, 100, 130)
    l, r = torch.from_numpy(left).to(DEV), torch.from_numpy(right).to(DEV)
    fs = HipOpticalFlowRAFT(None, None, iters=2)
    a = fs.compute_flow_sequence(l)
    b = fs.compute_flow_batch(gray_batch_to_rgb(l[:-1]), gray_batch_to_rgb(l[1:]))
    assert a.shape == (2, 2, 100, 130) and _epe(a, b) < 1e-3
    ds = HipDisparityRAFT(None, None, iters=2, config=RAFTStereoConfig.realtime())
    da = ds.compute_disparity_gray_batch(l, r)
    db = ds.compute_disparity_batch(gray_batch_to_rgb(l), gray_batch_to_rgb(r))
    assert da.shape == (3, 100, 130) and (da - db).abs().mean().item() < 1e-3


def test_fused_encoder_passes_match_stock_operators(vo):
    """instance-norm+ReLU, folded batch-norm+ReLU and add+ReLU against torch."""
    import torch.nn as nn
    from visual_odometry_amd import ops
    from visual_odometry_amd.raft.layers import BasicEncoder, fused_passes
    for shape in [(2, 5, 37, 41), (1, 64, 60, 80), (3, 8, 16, 16), (1, 3, 1, 7)]:
        x = _rand(*shape, seed=21, scale=2.0).to(DEV) + 0.5
        for relu in (True, False):
            want = nn.functional.instance_norm(x, eps=1e-5)
            want = torch.relu(want) if relu else want
            got = ops.instance_norm_relu_(x.clone(), 1e-5, relu)
            assert (got - want).abs().max().item() < 2e-5
        bn = nn.BatchNorm2d(shape[1]).to(DEV).eval()
        with torch.no_grad():
            bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.normal_(); bn.bias.normal_()
            want = torch.relu(bn(x))
            scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            got = ops.channel_affine_relu_(x.clone(), scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous(), True)
        assert (got - want).abs().max().item() < 2e-5
        y = _rand(*shape, seed=22).to(DEV)
        assert torch.equal(ops.add_relu_(x.clone(), y), torch.relu(x + y))
        # the residual join folded into the norm pass
        want = torch.relu(y + torch.relu(nn.functional.instance_norm(x, eps=1e-5)))
        assert (ops.instance_norm_relu_(x.clone(), 1e-5, True, skip=y) - want).abs().max().item() < 2e-5
        with torch.no_grad():
            want = torch.relu(y + torch.relu(bn(x)))
            got = ops.channel_affine_relu_(x.clone(), scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous(),
                                           True, skip=y)
        assert (got - want).abs().max().item() < 2e-5
        with pytest.raises(ValueError):
            ops.instance_norm_relu_(x.clone(), 1e-5, True, skip=y[:, :, :-1] if shape[2] > 1 else y[:, :-1])
    # a whole encoder, fused against stock, both norm kinds
    for kind in ("instance", "batch"):
        torch.manual_seed(3)
        enc = BasicEncoder(64, kind, downsample=2).to(DEV).eval()
        img = _rand(2, 3, 64, 96, seed=23).to(DEV)
        with torch.no_grad():
            a = enc(img)
            with fused_passes():
                b = enc(img)
        assert (a - b).abs().max().item() < 1e-4 * max(1.0, a.abs().max().item())


def test_fused_gru_steps_match_stock_operators(vo):
    from visual_odometry_amd.raft.layers import fused_passes
    from visual_odometry_amd.raft.update import BasicMultiUpdateBlock, BasicUpdateBlock
    torch.manual_seed(5)
    B, H, W = 2, 12, 20
    blk = BasicUpdateBlock().to(DEV).eval()
    net, inp = torch.tanh(_rand(B, 128, H, W, seed=30)).to(DEV), torch.relu(_rand(B, 128, H, W, seed=31)).to(DEV)
    corr, flow = _rand(B, 324, H, W, seed=32).to(DEV), _rand(B, 2, H, W, seed=33).to(DEV)
    with torch.no_grad():
        a = blk(net, inp, corr, flow, want_mask=True)
        with fused_passes():
            b = blk(net, inp, corr, flow, want_mask=True)
    for x, y in zip(a, b):
        assert (x - y).abs().max().item() < 1e-4 * max(1.0, x.abs().max().item())
    for layers, dims in ((2, [64, 64, 64]), (3, [128, 128, 128])):
        mb = BasicMultiUpdateBlock(dims, layers, 2).to(DEV).eval()
        nets = [torch.tanh(_rand(B, dims[2 - i], H >> i, W >> i, seed=40 + i)).to(DEV) for i in range(layers)]
        inps = [[_rand(B, dims[2 - i], H >> i, W >> i, seed=50 + 3 * i + k).to(DEV) for k in range(3)] for i in range(layers)]
        # context terms as channel slices of one tensor, as the model produces them
        inps = [list(torch.cat(t, dim=1).split(t[0].shape[1], dim=1)) for t in inps]
        corr1, flow1 = _rand(B, 36, H, W, seed=60).to(DEV), _rand(B, 2, H, W, seed=61).to(DEV)
        with torch.no_grad():
            a = mb([n.clone() for n in nets], inps, corr1, flow1, iter32=layers == 3, iter16=layers >= 2)
            with fused_passes():
                b = mb([n.clone() for n in nets], inps, corr1, flow1, iter32=layers == 3, iter16=layers >= 2)
        for x, y in zip(a[0], b[0]):
            assert (x - y).abs().max().item() < 1e-4
        assert (a[1] - b[1]).abs().max().item() < 1e-3 and (a[2] - b[2]).abs().max().item() < 1e-4


@pytest.mark.parametrize("B,C,X,H,W", [(2, 128, 256, 60, 80), (1, 64, 192, 24, 64), (2, 64, 64, 15, 20), (1, 128, 128, 9, 36)])
@pytest.mark.parametrize("ks", [(3, 3), (1, 5), (5, 1)])
def test_gru_step_with_gates_in_the_convolution_store(vo, B, C, X, H, W, ks):
    """vo_gru_conv_zr_bf16x3 + vo_gru_conv_q_bf16x3 (gate arithmetic in the convolutions' stores, r*h read by the q
    convolution in place of the buffer's first channels) against the stock operators of one GRU step: both
    block geometries, with and without context terms, the copy of the new state into the buffer slice."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    pad = (ks[0] // 2, ks[1] // 2)
    h = torch.tanh(_rand(B, C, H, W, seed=200)).to(DEV)
    x = _rand(B, X, H, W, seed=201).to(DEV)
    wz, wr, wq = ((_rand(C, C + X, *ks, seed=202 + i) / (ks[0] * ks[1] * (C + X)) ** 0.5).to(DEV) for i in range(3))
    bz, br, bq = (_rand(C, seed=205 + i).to(DEV) for i in range(3))
    ctx_all = (_rand(B, 3 * C, H, W, seed=208) * 0.5).to(DEV)
    for ctx in ((None, None, None), tuple(ctx_all.split(C, dim=1))):
        buf = torch.cat([h, x], dim=1).contiguous()
        hx = buf.double()
        add = lambda t: 0.0 if t is None else t.double()
        z = torch.sigmoid(Fn.conv2d(hx, wz.double(), bz.double(), 1, pad) + add(ctx[0]))
        r = torch.sigmoid(Fn.conv2d(hx, wr.double(), br.double(), 1, pad) + add(ctx[1]))
        q = torch.tanh(Fn.conv2d(torch.cat([r * h.double(), x.double()], dim=1), wq.double(), bq.double(), 1, pad) + add(ctx[2]))
        want = ((1 - z) * h.double() + z * q).float()
        Uzr = ops.gru_split_filter(torch.cat([wz, wr], dim=0).contiguous())
        Uq = ops.gru_split_filter(wq)
        zg, rh = ops.gru_conv_zr(buf, Uzr, torch.cat([bz, br]).contiguous(), ctx[0], ctx[1], h, ks)
        assert (zg - z.float()).abs().max().item() < 2e-5 and (rh - (r * h.double()).float()).abs().max().item() < 2e-5
        got = ops.gru_conv_q(buf, rh, Uq, bq, ctx[2], zg, h, ks, h_copy=buf[:, :C])
        assert (got - want).abs().max().item() < 5e-5
        assert torch.equal(buf[:, :C], got) and torch.equal(buf[:, C:], x)      # the state lands in the buffer, the inputs stay
    with pytest.raises(Exception):
        ops.gru_conv_zr(torch.cat([h, x], dim=1)[:, :, :, :W - 2].contiguous(), Uzr, None, None, None, h[:, :, :, :W - 2].contiguous(), ks)


def test_flow_update_block_with_folded_context_terms(vo):
    """RAFT's update block over three iterations with the persistent buffer: the context features' share of the
    six GRU gate convolutions is evaluated once and added in the convolutions' stores, the buffer holds
    [h | motion | flow] only; against the stock operators iteration by iteration."""
    from visual_odometry_amd.raft.layers import fused_passes
    from visual_odometry_amd.raft.update import BasicUpdateBlock
    torch.manual_seed(6)
    B, H, W = 1, 24, 32
    blk = BasicUpdateBlock().to(DEV).eval()
    net0, inp = torch.tanh(_rand(B, 128, H, W, seed=230)).to(DEV), torch.relu(_rand(B, 128, H, W, seed=231)).to(DEV)
    state, net_a, net_b = {}, net0.clone(), net0.clone()
    for it in range(3):
        corr, flow = _rand(B, 324, H, W, seed=232 + it).to(DEV), _rand(B, 2, H, W, seed=240 + it).to(DEV)
        with torch.no_grad():
            net_a, mask_a, delta_a = blk(net_a, inp, corr, flow, want_mask=True)
            with fused_passes():
                net_b, mask_b, delta_b = blk(net_b, inp, corr, flow, want_mask=True, state=state)
        assert state["fold"] == (128, 256) and state["buf"].shape[1] == 256
        for x, y in ((net_a, net_b), (mask_a, mask_b), (delta_a, delta_b)):
            assert (x - y).abs().max().item() < 2e-4 * max(1.0, x.abs().max().item())


@pytest.mark.parametrize("shape", [(2, 3, 30, 40), (1, 2, 7, 9), (2, 4, 1, 1), (1, 3, 16, 22), (1, 1, 5, 64)])
def test_avg_pool3x3s2_matches_avg_pool2d(vo, shape):
    """RAFT-Stereo's pool2x (3x3 / stride 2 / padding 1, padding counted) in float32 and float16
    storage: even and odd sizes, widths with and without the 16-byte paths."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(*shape, seed=77).to(DEV)
    want = Fn.avg_pool2d(x.double(), 3, stride=2, padding=1)
    got = ops.avg_pool3x3s2(x)
    assert got.shape == want.shape
    assert (got.double() - want).abs().max().item() < 1e-6
    got16 = ops.avg_pool3x3s2(x.half())
    assert got16.dtype == torch.float16
    assert (got16.double() - Fn.avg_pool2d(x.half().double(), 3, stride=2, padding=1)).abs().max().item() < 2e-3


@pytest.mark.parametrize("shape,size", [((2, 5, 6, 10), (12, 20)), ((1, 64, 60, 80), (120, 160)), ((1, 3, 7, 5), (9, 13)),
                                        ((1, 2, 1, 4), (1, 9)), ((2, 3, 8, 8), (8, 8)), ((1, 2, 3, 4), (6, 8)),
                                        ((2, 3, 30, 40), (60, 80)), ((1, 2, 5, 6), (7, 9 * 4)), ((1, 3, 9, 12), (11, 16)),
                                        ((1, 2, 4, 14), (9, 20))])
def test_resize_bilinear_matches_interpolate(vo, shape, size):
    """Enlargements by 1.5 or more with output rows in whole groups of four take the kernel that loads a window of
    four source columns per row (source widths of exactly four, windows clamped at the right edge, ratios at the
    limit of the window: 14 -> 20 is 0.684 and stays on the general kernel); everything else takes the general kernel."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(*shape, seed=70).to(DEV)
    got = ops.resize_bilinear(x, size)
    want = Fn.interpolate(x, size, mode="bilinear", align_corners=True)
    # the source coordinate scale*x may differ by one float ulp (fused vs separate multiply):
    # up to ~8e-6 of a pixel at these widths, times the local slope of the data
    assert got.shape == want.shape and (got - want).abs().max().item() < 1e-4
    half = ops.resize_bilinear(x.half(), size)
    want16 = Fn.interpolate(x.half().float(), size, mode="bilinear", align_corners=True)
    assert half.dtype == torch.float16 and (half.float() - want16).abs().max().item() < 4e-3


def test_fused_passes_in_float16_storage(vo):
    """The float16-storage variants of the fused passes against float32 math on the same
    (already rounded) inputs: only the final rounding to float16 may differ."""
    import torch.nn as nn
    from visual_odometry_amd import ops
    for shape in [(2, 6, 37, 40), (1, 64, 60, 80), (2, 3, 5, 7)]:
        x = (_rand(*shape, seed=41, scale=2.0) + 0.5).half().to(DEV)
        y = _rand(*shape, seed=42).half().to(DEV)
        xf, yf = x.float(), y.float()
        want = torch.relu(yf + torch.relu(nn.functional.instance_norm(xf, eps=1e-5)))
        got = ops.instance_norm_relu_(x.clone(), 1e-5, True, skip=y)
        assert got.dtype == torch.float16 and (got.float() - want).abs().max().item() < 4e-3
        scale, shift = _rand(shape[1], seed=43).to(DEV), _rand(shape[1], seed=44).to(DEV)
        want = torch.relu(xf * scale[None, :, None, None] + shift[None, :, None, None])
        got = ops.channel_affine_relu_(x.clone(), scale, shift, True)
        assert (got.float() - want).abs().max().item() < 1e-2 * max(1.0, want.abs().max().item()) * 0.1
        assert torch.equal(ops.add_relu_(x.clone(), y), torch.relu(xf + yf).half())
        up = ops.resize_bilinear(x, (shape[2] * 2, shape[3] * 2))
        want = nn.functional.interpolate(xf, (shape[2] * 2, shape[3] * 2), mode="bilinear", align_corners=True)
        assert up.dtype == torch.float16 and (up.float() - want).abs().max().item() < 4e-3
    with pytest.raises(ValueError):
        ops.add_relu_(x.clone(), y.float())
    with pytest.raises(ValueError):
        ops.instance_norm_relu_(x.double())


def test_mixed_precision_stays_within_the_flow_budget(vo):
    """float16 activations in both networks: flow and disparity stay within the 0.05 px
    mean end-point-error budget of the float32 path (the reference's precision)."""
    from visual_odometry_amd import synthetic as syn
    from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
    from visual_odometry_amd.raft.models import RAFTStereoConfig
    H, W, F = 128, 160, 4
    left, right = syn.stereo_sequence(5, F, H, W)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    outs = []
    for stereo_cfg in (RAFTStereoConfig.realtime(), RAFTStereoConfig()):
        res = []
        for mixed in (False, True):
            pipe = VisualOdometry(VOConfig(iters=6, stereo=stereo_cfg, mixed_precision=mixed), syn.left_K(0.25),
                                  syn.focal_length_px() * 0.25, syn.baseline(), np.ones((H, W), bool), seed=0)
            with torch.no_grad():
                res.append((pipe.flows(l), pipe.disparity_solver.compute_disparity_gray_batch(l, r)))
        (f32, d32), (f16, d16) = res
        assert f16.dtype == torch.float32 and d16.dtype == torch.float32
        epe = (f16 - f32).pow(2).sum(1).sqrt().mean().item()
        derr = (d16 - d32).abs().mean().item()
        assert epe < 0.05 and derr < 0.05, (epe, derr)
        assert epe > 0 and derr > 0            # the switch did change the arithmetic


WINO_TOL = {2: 2e-5, 4: 1e-4}      # relative to the largest output value


@pytest.mark.parametrize("N,C,H,W,K", [(2, 8, 16, 16, 64), (1, 16, 37, 53, 40), (2, 64, 60, 80, 126), (1, 24, 17, 16, 32),
                                       (3, 128, 33, 47, 192), (1, 8, 1, 1, 33), (2, 96, 30, 40, 96), (1, 40, 70, 100, 72)])
@pytest.mark.parametrize("tile", [2, 4])
def test_winograd_conv3x3_matches_direct_convolution(vo, N, C, H, W, K, tile):
    """The Winograd/MFMA 3x3 kernels, F(2x2,3x3) and F(4x4,3x3), against the float64 direct
    convolution: ragged image sizes (partial blocks), output channels that do not fill a
    64-block, and run-to-run identical results.  Tolerance: 2e-5 / 1e-4 of the largest output
    (the larger tile's transforms multiply by up to 8, measured error 1e-5)."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=50).to(DEV)
    w = (_rand(K, C, 3, 3, seed=51) / (3 * C ** 0.5)).to(DEV)
    U = ops.winograd_filter(w, tile)
    got = ops.conv3x3_winograd(x, U, K)
    want = Fn.conv2d(x.double(), w.double(), None, 1, 1)
    assert got.shape == want.shape
    assert (got.double() - want).abs().max().item() < WINO_TOL[tile] * max(1.0, want.abs().max().item())
    assert torch.equal(got, ops.conv3x3_winograd(x, U, K))
    with pytest.raises(ValueError):
        ops.winograd_filter(_rand(8, 6, 3, 3, seed=1).to(DEV))
    with pytest.raises(ValueError):
        ops.conv3x3_winograd(x, U[:-1].contiguous(), K)


@pytest.mark.parametrize("tile", [2, 4, 0])
def test_networks_agree_with_and_without_the_winograd_kernel(vo, tile):
    """Flow and disparity with every 3x3 layer on MIOpen against the Winograd kernels: forced
    F(2x2,3x3), forced F(4x4,3x3) and the per-shape choice.  Through three update iterations the
    larger tile's rounding shows as at most a few 1e-4 px."""
    from visual_odometry_amd.raft import layers
    from visual_odometry_amd.raft.models import RAFTFlow, RAFTStereo, RAFTStereoConfig
    torch.manual_seed(11)
    flow, stereo = RAFTFlow().to(DEV).eval(), RAFTStereo(RAFTStereoConfig.realtime()).to(DEV).eval()
    a, b = (_rand(1, 3, 96, 128, seed=60) * 60 + 120).to(DEV), (_rand(1, 3, 96, 128, seed=61) * 60 + 120).to(DEV)
    with layers.arithmetic(hand_kernels=False):
        f0, d0 = flow(a, b, iters=3)[1], stereo(a, b, iters=3)[1]
    with layers.arithmetic(winograd_tile=tile):
        f1, d1 = flow(a, b, iters=3)[1], stereo(a, b, iters=3)[1]
    tol = 1e-4 if tile == 2 else 2e-3
    assert (f1 - f0).abs().max().item() < tol and (d1 - d0).abs().max().item() < tol


@pytest.mark.parametrize("tile", [2, 4])
def test_winograd_fused_epilogue(vo, tile):
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    N, C, H, W, K = 2, 16, 21, 28, 40
    x, w = _rand(N, C, H, W, seed=70).to(DEV), (_rand(K, C, 3, 3, seed=71) / 12).to(DEV)
    scale, shift, skip = _rand(K, seed=72).to(DEV), _rand(K, seed=73).to(DEV), _rand(N, K, H, W, seed=74).to(DEV)
    U = ops.winograd_filter(w, tile)
    y = Fn.conv2d(x, w, None, 1, 1)
    b = lambda v: v[None, :, None, None]
    cases = [((None, shift, False, None), y + b(shift)), ((None, shift, True, None), torch.relu(y + b(shift))),
             ((scale, shift, True, None), torch.relu(y * b(scale) + b(shift))),
             ((scale, shift, True, skip), torch.relu(skip + torch.relu(y * b(scale) + b(shift)))),
             ((None, None, True, skip), torch.relu(skip + torch.relu(y)))]
    for args, want in cases:
        got = ops.conv3x3_winograd(x, U, K, *args)
        assert (got - want).abs().max().item() < WINO_TOL[tile] * max(1.0, want.abs().max().item())
    with pytest.raises(ValueError):
        ops.conv3x3_winograd(x, U, K, scale, None)
    with pytest.raises(ValueError):
        ops.conv3x3_winograd(x, U, K, None, shift[:-1].contiguous())


def test_small_output_conv_as_taps_plus_shift_add(vo):
    import torch.nn as nn
    from visual_odometry_amd.raft.layers import conv_plain, fused_passes
    torch.manual_seed(2)
    for (N, C, H, W, K) in [(2, 32, 13, 300, 2), (1, 256, 60, 80, 2), (2, 8, 5, 7, 3)]:
        conv = nn.Conv2d(C, K, 3, padding=1).to(DEV)
        x = _rand(N, C, H, W, seed=80).to(DEV)
        with torch.no_grad():
            want = conv(x)
            with fused_passes():
                got = conv_plain(x, conv)
        assert got.shape == want.shape and (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("tile", [2, 4])
def test_instance_norm_from_convolution_statistics(vo, tile):
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    for (N, C, H, W, K) in [(2, 16, 21, 28, 40), (3, 64, 60, 80, 64), (1, 8, 16, 16, 33)]:
        x, w = _rand(N, C, H, W, seed=90).to(DEV), (_rand(K, C, 3, 3, seed=91) / 12).to(DEV)
        skip = _rand(N, K, H, W, seed=92).to(DEV)
        U = ops.winograd_filter(w, tile)
        stats = torch.zeros((N, K, 2), device=DEV, dtype=torch.float64)
        y = ops.conv3x3_winograd(x, U, K, stats=stats)
        ref_y = Fn.conv2d(x.double(), w.double(), None, 1, 1)
        want_stats = torch.stack([ref_y.sum((2, 3)), (ref_y * ref_y).sum((2, 3))], dim=2)
        assert (stats - want_stats).abs().max().item() < 1e-4 * max(1.0, want_stats.abs().max().item())
        want = torch.relu(skip + torch.relu(Fn.instance_norm(ref_y.float(), eps=1e-5)))
        got = ops.instance_norm_apply_(y, stats, 1e-5, True, skip)
        assert (got - want).abs().max().item() < (5e-5 if tile == 2 else 3e-4)
    # a plane whose mean is a hundred times its spread: the variance formed from the kernel's sums still matches
    # the variance of the kernel's own output (cross-lane sums in float64)
    x = (50.0 + 0.05 * _rand(1, 16, 64, 96, seed=93)).to(DEV)
    w = (_rand(32, 16, 3, 3, seed=94).abs() / 50).to(DEV)
    stats = torch.zeros((1, 32, 2), device=DEV, dtype=torch.float64)
    full = ops.conv3x3_winograd(x, ops.winograd_filter(w, tile), 32, stats=stats).double()
    hw = full.shape[2] * full.shape[3]
    mean_k, var_k = stats[0, :, 0] / hw, stats[0, :, 1] / hw - (stats[0, :, 0] / hw) ** 2
    mean_o, var_o = full.mean((2, 3))[0], full.var((2, 3), unbiased=False)[0]
    assert (mean_k - mean_o).abs().max().item() < 1e-7 * mean_o.abs().max().item()
    assert ((var_k - var_o).abs() / var_o).max().item() < 1e-3
    with pytest.raises(ValueError):
        ops.conv3x3_winograd(x, U, K, None, skip[0, :, 0, 0].contiguous(), False, None, None, stats)


@pytest.mark.parametrize("tile", [2, 4])
def test_winograd_normalises_its_input_on_load(vo, tile):
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    from visual_odometry_amd.raft import layers
    for (N, C, H, W, K) in [(2, 16, 21, 28, 40), (2, 64, 33, 47, 64)]:
        z = (_rand(N, C, H, W, seed=95, scale=2.0) + 0.7).to(DEV)           # a raw convolution output
        w = (_rand(K, C, 3, 3, seed=96) / 12).to(DEV)
        st = torch.stack([z.double().sum((2, 3)), (z.double() ** 2).sum((2, 3))], dim=2).contiguous()
        got = ops.conv3x3_winograd(z, ops.winograd_filter(w, tile), K, in_stats=st, in_eps=1e-5)
        want = Fn.conv2d(torch.relu(Fn.instance_norm(z, eps=1e-5)), w, None, 1, 1)
        assert (got - want).abs().max().item() < 1.5 * WINO_TOL[tile] * max(1.0, want.abs().max().item())
    # a residual block end to end, chained path against the stock operators
    from visual_odometry_amd.raft.layers import ResidualBlock, fused_passes
    torch.manual_seed(4)
    blk = ResidualBlock(64, 64, "instance").to(DEV).eval()
    x = torch.relu(_rand(2, 64, 37, 52, seed=97)).to(DEV)
    with layers.arithmetic(winograd_tile=tile), torch.no_grad():
        a = blk(x)
        with fused_passes():
            b = blk(x)
    assert (a - b).abs().max().item() < (1e-4 if tile == 2 else 5e-4) * max(1.0, a.abs().max().item())


@pytest.mark.parametrize("tile", [2, 4])
def test_kernels_write_channel_slices_in_place(vo, tile):
    """Winograd convolution and bilinear resize writing straight into a channel slice of a
    wider tensor (the consumer's concatenated input): neighbours untouched, values equal."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    N, C, H, W, K = 2, 16, 18, 24, 40
    x, w = _rand(N, C, H, W, seed=100).to(DEV), (_rand(K, C, 3, 3, seed=101) / 12).to(DEV)
    bias = _rand(K, seed=102).to(DEV)
    wide = torch.full((N, 7 + K + 5, H, W), 3.25, device=DEV)
    got = ops.conv3x3_winograd(x, ops.winograd_filter(w, tile), K, None, bias, True, out=wide[:, 7:7 + K])
    want = torch.relu(Fn.conv2d(x, w, bias, 1, 1))
    assert got.data_ptr() == wide[:, 7:7 + K].data_ptr()
    assert (wide[:, 7:7 + K] - want).abs().max().item() < WINO_TOL[tile] * max(1.0, want.abs().max().item())
    assert torch.all(wide[:, :7] == 3.25) and torch.all(wide[:, 7 + K:] == 3.25)
    small = _rand(N, 6, 9, 12, seed=103).to(DEV)
    wide = torch.full((N, 10, H, W), -1.5, device=DEV)
    ops.resize_bilinear(small, (H, W), out=wide[:, 2:8])
    want = Fn.interpolate(small, (H, W), mode="bilinear", align_corners=True)
    assert (wide[:, 2:8] - want).abs().max().item() < 1e-4
    assert torch.all(wide[:, :2] == -1.5) and torch.all(wide[:, 8:] == -1.5)
    with pytest.raises(ValueError):
        ops.resize_bilinear(small, (H, W), out=wide[:, 2:7])
    with pytest.raises(ValueError):
        ops.conv3x3_winograd(x, ops.winograd_filter(w, tile), K, out=wide.permute(0, 1, 3, 2)[:, :K])
    # the 3x3 stride-2 average into a slice (vo_avg_pool3x3s2_into): float32 and float16, rows of 12 (16-byte stores)
    # and of 7 pixels (element stores), bit for bit what the dense call returns
    for (h, w) in ((18, 24), (9, 13)):
        for dt in (torch.float32, torch.float16):
            big = _rand(N, 5, h, w, seed=104).to(DEV).to(dt)
            dense = ops.avg_pool3x3s2(big)
            wide = torch.full((N, 9, (h - 1) // 2 + 1, (w - 1) // 2 + 1), 0.75, device=DEV, dtype=dt)
            got = ops.avg_pool3x3s2(big, out=wide[:, 3:8])
            assert got.data_ptr() == wide[:, 3:8].data_ptr() and torch.equal(wide[:, 3:8], dense)
            assert torch.all(wide[:, :3] == 0.75) and torch.all(wide[:, 8:] == 0.75)
            want = Fn.avg_pool2d(big.double(), 3, stride=2, padding=1)
            assert (dense.double() - want).abs().max().item() < (1e-6 if dt == torch.float32 else 2e-3)
    with pytest.raises(ValueError):
        ops.avg_pool3x3s2(big, out=wide[:, 3:7])


SPLIT_TOL = 2e-5                   # relative to the largest output value (measured 4-6e-6)


@pytest.mark.parametrize("N,C,H,W,K", [(2, 16, 16, 32, 64), (1, 16, 37, 54, 40), (2, 64, 60, 80, 126), (1, 32, 17, 16, 32),
                                       (3, 128, 33, 46, 192), (1, 16, 1, 2, 33), (2, 96, 30, 40, 96), (1, 48, 70, 100, 72),
                                       (1, 256, 15, 20, 128), (5, 64, 120, 160, 64), (2, 32, 50, 74, 33), (1, 16, 63, 14, 64)])
def test_split_bf16_conv3x3_matches_direct_convolution(vo, N, C, H, W, K):
    """The split-bf16 direct 3x3 kernel (each float32 product from three bf16 matrix-core terms, float32
    accumulation) against the float64 direct convolution: ragged image sizes, one-pixel-high and
    two-pixel-wide maps, output channels that do not fill a 64-block (half blocks included), several
    chunks of input channels, more tiles than workgroup slots, maps that take the 32x16 block geometry
    (60x80, 30x40, 50x74, 63x14) as well as the 16x32 one, run-to-run identical results.
    Tolerance 2e-5 of the largest output, the bar of the float32 F(2x2,3x3) kernel."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=50).to(DEV)
    w = (_rand(K, C, 3, 3, seed=51) / (3 * C ** 0.5)).to(DEV)
    U = ops.conv3x3_split_filter(w)
    got = ops.conv3x3_winograd(x, U, K)
    want = Fn.conv2d(x.double(), w.double(), None, 1, 1)
    assert got.shape == want.shape
    assert (got.double() - want).abs().max().item() < SPLIT_TOL * max(1.0, want.abs().max().item())
    assert torch.equal(got, ops.conv3x3_winograd(x, U, K))
    # inputs spanning many binades: the split keeps 16 significant bits of EVERY product
    xs = x * torch.exp2(torch.randint(-12, 12, x.shape, device=DEV).float())
    gots = ops.conv3x3_winograd(xs, U, K)
    wants = Fn.conv2d(xs.double(), w.double(), None, 1, 1)
    assert (gots.double() - wants).abs().max().item() < SPLIT_TOL * max(1.0, wants.abs().max().item())
    with pytest.raises(ValueError):
        ops.conv3x3_split_filter(_rand(8, 24, 3, 3, seed=1).to(DEV))           # channels not a multiple of 16
    with pytest.raises(ValueError):
        ops.conv3x3_winograd(_rand(1, C, 8, 9, seed=2).to(DEV), U, K)          # odd width


def test_split_bf16_conv3x3_epilogue_statistics_and_slices(vo):
    """Fused epilogue (scale / shift / ReLU / residual join), plane statistics for a following instance
    norm, normalise-on-load of the input and channel-slice output of the split-bf16 kernel, each against
    stock operators; then a residual block with instance norm end to end on it."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    from visual_odometry_amd.raft import layers
    b = lambda v: v[None, :, None, None]
    for (N, C, H, W, K) in [(2, 32, 50, 74, 40), (2, 32, 21, 28, 40)]:      # 32x16 blocks, then 16x32 blocks
        x, w = _rand(N, C, H, W, seed=70).to(DEV), (_rand(K, C, 3, 3, seed=71) / 12).to(DEV)
        scale, shift, skip = _rand(K, seed=72).to(DEV), _rand(K, seed=73).to(DEV), _rand(N, K, H, W, seed=74).to(DEV)
        U = ops.conv3x3_split_filter(w)
        y = Fn.conv2d(x.double(), w.double(), None, 1, 1).float()
        cases = [((None, shift, False, None), y + b(shift)), ((None, shift, True, None), torch.relu(y + b(shift))),
                 ((scale, shift, True, None), torch.relu(y * b(scale) + b(shift))),
                 ((scale, shift, True, skip), torch.relu(skip + torch.relu(y * b(scale) + b(shift)))),
                 ((None, None, True, skip), torch.relu(skip + torch.relu(y)))]
        for args, want in cases:
            got = ops.conv3x3_winograd(x, U, K, *args)
            assert (got - want).abs().max().item() < SPLIT_TOL * max(1.0, want.abs().max().item())
    wide = torch.full((N, 7 + K + 5, H, W), 3.25, device=DEV)
    got = ops.conv3x3_winograd(x, U, K, None, shift, True, out=wide[:, 7:7 + K])
    assert got.data_ptr() == wide[:, 7:7 + K].data_ptr()
    assert (wide[:, 7:7 + K] - torch.relu(y + b(shift))).abs().max().item() < SPLIT_TOL * max(1.0, y.abs().max().item())
    assert torch.all(wide[:, :7] == 3.25) and torch.all(wide[:, 7 + K:] == 3.25)
    # statistics of the raw output -> one-pass instance norm (+ ReLU + residual join)
    for (N, C, H, W, K) in [(2, 16, 21, 28, 40), (3, 64, 60, 80, 64), (1, 16, 16, 16, 33), (2, 96, 35, 66, 96), (2, 16, 50, 74, 40)]:
        x, w = _rand(N, C, H, W, seed=90).to(DEV), (_rand(K, C, 3, 3, seed=91) / 12).to(DEV)
        skip = _rand(N, K, H, W, seed=92).to(DEV)
        U = ops.conv3x3_split_filter(w)
        stats = torch.zeros((N, K, 2), device=DEV, dtype=torch.float64)
        yy = ops.conv3x3_winograd(x, U, K, stats=stats)
        ref_y = Fn.conv2d(x.double(), w.double(), None, 1, 1)
        want_stats = torch.stack([ref_y.sum((2, 3)), (ref_y * ref_y).sum((2, 3))], dim=2)
        assert (stats - want_stats).abs().max().item() < 1e-4 * max(1.0, want_stats.abs().max().item())
        want = torch.relu(skip + torch.relu(Fn.instance_norm(ref_y.float(), eps=1e-5)))
        got = ops.instance_norm_apply_(yy, stats, 1e-5, True, skip)
        assert (got - want).abs().max().item() < 5e-5
    # a plane whose mean is a hundred times its spread (a near-constant image region): the variance
    # formed from the kernel's sums still matches the variance of the kernel's own output
    x = (50.0 + 0.05 * _rand(1, 16, 64, 96, seed=93)).to(DEV)
    w = (_rand(32, 16, 3, 3, seed=94).abs() / 50).to(DEV)
    stats = torch.zeros((1, 32, 2), device=DEV, dtype=torch.float64)
    full = ops.conv3x3_winograd(x, ops.conv3x3_split_filter(w), 32, stats=stats).double()
    hw = full.shape[2] * full.shape[3]
    mean_k, var_k = stats[0, :, 0] / hw, stats[0, :, 1] / hw - (stats[0, :, 0] / hw) ** 2
    mean_o, var_o = full.mean((2, 3))[0], full.var((2, 3), unbiased=False)[0]
    assert (mean_k - mean_o).abs().max().item() < 1e-7 * mean_o.abs().max().item()
    assert ((var_k - var_o).abs() / var_o).max().item() < 1e-3 and (mean_o.abs() / var_o.sqrt()).min().item() > 3
    # the input normalised + ReLU-ed while it is staged
    for (N, C, H, W, K) in [(2, 16, 21, 28, 40), (2, 64, 33, 46, 64), (1, 128, 30, 40, 128), (1, 256, 30, 40, 64), (2, 48, 50, 74, 40)]:
        z = (_rand(N, C, H, W, seed=95, scale=2.0) + 0.7).to(DEV)
        w = (_rand(K, C, 3, 3, seed=96) / 12).to(DEV)
        st = torch.stack([z.double().sum((2, 3)), (z.double() ** 2).sum((2, 3))], dim=2).contiguous()
        got = ops.conv3x3_winograd(z, ops.conv3x3_split_filter(w), K, in_stats=st, in_eps=1e-5)
        want = Fn.conv2d(torch.relu(Fn.instance_norm(z.double(), eps=1e-5)), w.double(), None, 1, 1)
        assert (got.double() - want).abs().max().item() < 1.5 * SPLIT_TOL * max(1.0, want.abs().max().item())
    from visual_odometry_amd.raft.layers import ResidualBlock, fused_passes
    torch.manual_seed(4)
    blk = ResidualBlock(64, 64, "instance").to(DEV).eval()
    xb = torch.relu(_rand(2, 64, 37, 52, seed=97)).to(DEV)
    assert layers._winograd_tile_for(37, 52, 64) == 1               # the block does run on the split kernel
    with torch.no_grad():
        a = blk(xb)
        with fused_passes():
            bb = blk(xb)
    assert (a - bb).abs().max().item() < 1e-4 * max(1.0, a.abs().max().item())


@pytest.mark.parametrize("N,C,H,W,K", [(1, 16, 5, 8, 64), (2, 32, 33, 22, 40), (2, 64, 60, 80, 126), (1, 16, 1, 2, 33),
                                       (2, 384, 30, 40, 256), (1, 48, 70, 4, 32), (3, 384, 60, 80, 128), (2, 32, 50, 74, 33)])
@pytest.mark.parametrize("vertical", [False, True])
def test_split_bf16_conv5_matches_direct_convolution(vo, N, C, H, W, K, vertical):
    """1x5 / 5x1 GRU gate convolutions on the split-bf16 direct kernel against the float64 direct
    convolution (ragged sizes, partial channel blocks, run-to-run identical), the fused bias + ReLU and
    a channel-slice output."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=110).to(DEV)
    shape = (K, C, 5, 1) if vertical else (K, C, 1, 5)
    w = (_rand(*shape, seed=111) / (5 * C) ** 0.5).to(DEV)
    U = ops.conv5_split_filter(w)
    got = ops.conv5_split(x, U, K, vertical)
    want = Fn.conv2d(x.double(), w.double(), None, 1, (2, 0) if vertical else (0, 2))
    assert got.shape == want.shape
    assert (got.double() - want).abs().max().item() < SPLIT_TOL * max(1.0, want.abs().max().item())
    assert torch.equal(got, ops.conv5_split(x, U, K, vertical))
    bias = _rand(K, seed=112).to(DEV)
    wide = torch.full((N, 3 + K + 2, H, W), 1.75, device=DEV)
    ops.conv5_split(x, U, K, vertical, None, bias, True, out=wide[:, 3:3 + K])
    ref = torch.relu(want.float() + bias[None, :, None, None])
    assert (wide[:, 3:3 + K] - ref).abs().max().item() < SPLIT_TOL * max(1.0, want.abs().max().item())
    assert torch.all(wide[:, :3] == 1.75) and torch.all(wide[:, 3 + K:] == 1.75)
    with pytest.raises(ValueError):
        ops.conv5_split_filter(_rand(8, 24, 1, 5, seed=1).to(DEV))


@pytest.mark.parametrize("N,C,H,W,K", [(1, 16, 5, 8, 64), (2, 16, 33, 70, 40), (2, 64, 9, 132, 100), (1, 16, 1, 2, 3),
                                       (2, 64, 60, 80, 96), (1, 96, 61, 62, 128), (3, 128, 120, 160, 128)])
def test_split_bf16_conv3x3s2_matches_the_stock_convolution(vo, N, C, H, W, K):
    """3x3 stride-2 convolution on the split-bf16 direct kernel (even and odd columns of the staged patch
    stored apart) against the float64 stock convolution: odd and even heights, ragged widths, the fused
    scale / shift / ReLU, and the plane statistics for a following instance norm."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=120).to(DEV)
    w = (_rand(K, C, 3, 3, seed=121) / (3 * C ** 0.5)).to(DEV)
    U = ops.conv3x3_split_filter(w)
    got = ops.conv3x3s2_split(x, U, K)
    want = Fn.conv2d(x.double(), w.double(), None, 2, 1)
    assert got.shape == want.shape
    tol = SPLIT_TOL * max(1.0, want.abs().max().item())
    assert (got.double() - want).abs().max().item() < tol
    assert torch.equal(got, ops.conv3x3s2_split(x, U, K))
    scale, shift = _rand(K, seed=122).to(DEV), _rand(K, seed=123).to(DEV)
    got = ops.conv3x3s2_split(x, U, K, scale, shift, True)
    ref = torch.relu(want.float() * scale[None, :, None, None] + shift[None, :, None, None])
    assert (got - ref).abs().max().item() < 2 * tol
    stats = torch.zeros((N, K, 2), device=DEV, dtype=torch.float64)
    raw = ops.conv3x3s2_split(x, U, K, stats=stats)
    want_stats = torch.stack([want.sum((2, 3)), (want * want).sum((2, 3))], dim=2)
    assert (stats - want_stats).abs().max().item() < 1e-4 * max(1.0, want_stats.abs().max().item())
    if want.shape[2] * want.shape[3] > 1:                       # torch refuses to normalise a one-pixel plane
        got = ops.instance_norm_apply_(raw, stats, 1e-5, True)
        assert (got - torch.relu(Fn.instance_norm(want.float(), eps=1e-5))).abs().max().item() < 5e-5


@pytest.mark.parametrize("N,C,H,W,K", [(1, 1, 5, 7, 64), (2, 2, 33, 70, 40), (2, 1, 9, 130, 96), (1, 1, 2, 3, 8),
                                       (2, 1, 96, 128, 64), (1, 2, 60, 80, 128), (3, 1, 61, 191, 64)])
@pytest.mark.parametrize("stride", [1, 2])
def test_split_bf16_conv7x7_matches_the_stock_convolution(vo, N, C, H, W, K, stride):
    """The one- and two-channel 7x7 kernel (stems on a gray frame, the motion encoders' flow branch) against
    the float64 stock convolution: rows not a multiple of four, widths not a multiple of 64, partial
    channel blocks, the fused scale / shift / ReLU, and the plane statistics a workgroup accumulates over
    whole rows."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=130).to(DEV)
    w = (_rand(K, C, 7, 7, seed=131) / (7 * C ** 0.5)).to(DEV)
    A = ops.conv7x7_split_filter(w)
    got = ops.conv7x7_split(x, A, K, stride=stride)
    want = Fn.conv2d(x.double(), w.double(), None, stride, 3)
    assert got.shape == want.shape
    tol = SPLIT_TOL * max(1.0, want.abs().max().item())
    assert (got.double() - want).abs().max().item() < tol
    assert torch.equal(got, ops.conv7x7_split(x, A, K, stride=stride))
    scale, shift = _rand(K, seed=132).to(DEV), _rand(K, seed=133).to(DEV)
    got = ops.conv7x7_split(x, A, K, scale, shift, True, stride)
    ref = torch.relu(want.float() * scale[None, :, None, None] + shift[None, :, None, None])
    assert (got - ref).abs().max().item() < 2 * tol
    got = ops.conv7x7_split(x, A, K, None, shift, False, stride)
    assert (got - (want.float() + shift[None, :, None, None])).abs().max().item() < 2 * tol
    stats = torch.zeros((N, K, 2), device=DEV, dtype=torch.float64)
    raw = ops.conv7x7_split(x, A, K, stride=stride, stats=stats)
    want_stats = torch.stack([want.sum((2, 3)), (want * want).sum((2, 3))], dim=2)
    assert (stats - want_stats).abs().max().item() < 1e-4 * max(1.0, want_stats.abs().max().item())
    if want.shape[2] * want.shape[3] > 1:
        got = ops.instance_norm_apply_(raw, stats, 1e-5, True)
        assert (got - torch.relu(Fn.instance_norm(want.float(), eps=1e-5))).abs().max().item() < 5e-5
    with pytest.raises(ValueError):
        ops.conv7x7_split_filter(_rand(8, 3, 7, 7, seed=1).to(DEV))
    with pytest.raises(ValueError):
        ops.conv7x7_split(x, A, K, stride=stride, stats=stats, relu=True)


@pytest.mark.parametrize("N,C,H,W,K", [(1, 16, 5, 8, 64), (2, 16, 33, 70, 40), (2, 64, 9, 132, 100), (1, 16, 1, 2, 3),
                                       (2, 64, 60, 80, 96), (1, 96, 61, 62, 128)])
def test_split_bf16_conv1x1s2_matches_the_stock_convolution(vo, N, C, H, W, K):
    """The projection shortcuts' 1x1 stride-2 convolution on the split-bf16 template (it stages only the rows
    the stride lands on) against the float64 stock convolution, with epilogue and plane statistics."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=140).to(DEV)
    w = (_rand(K, C, 1, 1, seed=141) / C ** 0.5).to(DEV)
    A = ops.conv1x1_split_filter(w)
    got = ops.conv1x1s2_split(x, A, K)
    want = Fn.conv2d(x.double(), w.double(), None, 2, 0)
    assert got.shape == want.shape
    tol = SPLIT_TOL * max(1.0, want.abs().max().item())
    assert (got.double() - want).abs().max().item() < tol
    scale, shift = _rand(K, seed=142).to(DEV), _rand(K, seed=143).to(DEV)
    got = ops.conv1x1s2_split(x, A, K, scale, shift, True)
    ref = torch.relu(want.float() * scale[None, :, None, None] + shift[None, :, None, None])
    assert (got - ref).abs().max().item() < 2 * tol
    stats = torch.zeros((N, K, 2), device=DEV, dtype=torch.float64)
    ops.conv1x1s2_split(x, A, K, stats=stats)
    want_stats = torch.stack([want.sum((2, 3)), (want * want).sum((2, 3))], dim=2)
    assert (stats - want_stats).abs().max().item() < 1e-4 * max(1.0, want_stats.abs().max().item())


@pytest.mark.parametrize("N,C,H,W,K", [(2, 64, 48, 64, 96), (1, 96, 30, 40, 128), (2, 64, 37, 24, 64), (1, 128, 16, 48, 160)])
def test_split_bf16_downsampling_pair_in_one_pass(vo, N, C, H, W, K):
    """vo_conv3x3s2_shortcut_bf16x3 against its two single kernels (bit for bit: the same products in the same
    order) and against the float64 stock convolutions: with plane sums for both outputs (the instance-norm
    blocks), and with a scale / shift / ReLU store for the first and scale / shift for the second (the folded
    batch norms); partial channel blocks and odd heights."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=180).to(DEV)
    w3 = (_rand(K, C, 3, 3, seed=181) / (3 * C ** 0.5)).to(DEV)
    w1 = (_rand(K, C, 1, 1, seed=182) / C ** 0.5).to(DEV)
    A3, A1 = ops.conv3x3_split_filter(w3), ops.conv1x1_split_filter(w1)
    st3, st1 = (torch.zeros((N, K, 2), device=DEV, dtype=torch.float64) for _ in range(2))
    y, sc = ops.conv3x3s2_shortcut_split(x, A3, A1, K, stats=st3, stats2=st1)
    r3, r1 = (torch.zeros((N, K, 2), device=DEV, dtype=torch.float64) for _ in range(2))
    assert torch.equal(y, ops.conv3x3s2_split(x, A3, K, stats=r3))
    assert torch.equal(sc, ops.conv1x1s2_split(x, A1, K, stats=r1))
    want3, want1 = Fn.conv2d(x.double(), w3.double(), None, 2, 1), Fn.conv2d(x.double(), w1.double(), None, 2, 0)
    assert (y.double() - want3).abs().max().item() < SPLIT_TOL * max(1.0, want3.abs().max().item())
    assert (sc.double() - want1).abs().max().item() < SPLIT_TOL * max(1.0, want1.abs().max().item())
    for st, want in ((st3, want3), (st1, want1)):
        sums = torch.stack([want.sum((2, 3)), (want * want).sum((2, 3))], dim=2)
        assert (st - sums).abs().max().item() < 1e-4 * max(1.0, sums.abs().max().item())
    s3, b3, s1, b1 = (_rand(K, seed=183 + i).to(DEV) for i in range(4))
    y, sc = ops.conv3x3s2_shortcut_split(x, A3, A1, K, s3, b3, True, None, s1, b1, False, None)
    assert torch.equal(y, ops.conv3x3s2_split(x, A3, K, s3, b3, True))
    assert torch.equal(sc, ops.conv1x1s2_split(x, A1, K, s1, b1, False))
    ref = torch.relu(want3.float() * s3[None, :, None, None] + b3[None, :, None, None])
    assert (y - ref).abs().max().item() < 4 * SPLIT_TOL * max(1.0, ref.abs().max().item())
    with pytest.raises(ValueError):
        ops.conv3x3s2_shortcut_split(x[:, :, :, :W - 2].contiguous() if (W - 2) // 2 % 4 else x[:, :, :, :W - 4].contiguous(), A3, A1, K)


@pytest.mark.parametrize("N,C,H,W,K", [(1, 64, 5, 8, 64), (2, 336, 60, 80, 256), (2, 128, 33, 70, 100), (1, 256, 15, 20, 576),
                                       (2, 64, 9, 132, 144)])
def test_split_bf16_conv1x1_matches_the_stock_convolution(vo, N, C, H, W, K):
    """The many-channel 1x1 layers on the split-bf16 template against the float64 stock convolution: both block
    geometries, partial channel blocks, fused bias + ReLU, channel-slice output; and a layer whose input was
    padded with zero planes (the correlation window: 324 -> 336 channels) through its zero-extended twin."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    from visual_odometry_amd.raft import layers
    x = _rand(N, C, H, W, seed=170).to(DEV)
    w = (_rand(K, C, 1, 1, seed=171) / C ** 0.5).to(DEV)
    bias = _rand(K, seed=172).to(DEV)
    A = ops.conv1x1_split_filter(w)
    want = Fn.conv2d(x.double(), w.double(), None)
    tol = SPLIT_TOL * max(1.0, want.abs().max().item())
    assert (ops.conv1x1_split(x, A, K).double() - want).abs().max().item() < tol
    wide = torch.full((N, 3 + K + 2, H, W), 1.75, device=DEV)
    ops.conv1x1_split(x, A, K, None, bias, True, out=wide[:, 3:3 + K])
    ref = torch.relu(want.float() + bias[None, :, None, None])
    assert (wide[:, 3:3 + K] - ref).abs().max().item() < 2 * tol
    assert torch.all(wide[:, :3] == 1.75) and torch.all(wide[:, 3 + K:] == 1.75)
    # the same layer written pixel-major (N,H,W,K), the correlation kernels' input layout
    pm = ops.conv1x1_pixel_major_split(x, A, K, None, bias, False)
    assert pm.shape == (N, H, W, K)
    assert (pm.permute(0, 3, 1, 2) - (want.float() + bias[None, :, None, None])).abs().max().item() < 2 * tol
    if K % 32 == 0:
        # ... and with its rows leaving as bf16 pieces: the bits of vo_split_feature_rows over the float32 rows
        sr = ops.conv1x1_pixel_major_split(x, A, K, None, bias, False, split_rows=True)
        assert torch.equal(sr.view(torch.int16), ops.split_feature_rows(pm, 32).view(torch.int16))
    if C == 336:
        conv = torch.nn.Conv2d(324, K, 1).to(DEV).eval()
        xp = x.clone(); xp[:, 324:] = 0.0
        with torch.no_grad():
            want2 = torch.relu(conv(xp[:, :324].contiguous()))
            with layers.fused_passes():
                got2 = layers.conv_act(xp, layers.padded_input_twin(conv, 336))
        assert (got2 - want2).abs().max().item() < 1e-4 * max(1.0, want2.abs().max().item())


def test_streaming_conv1x1_equals_the_template_kernel_bit_for_bit(vo):
    """Layers with at least 64 output channels in multiples of 8 and planes of whole groups of four pixels run on the kernel that
    keeps the weights in LDS (bf16 pieces); ``use_template`` keeps them on the convolution template.  Same weights image,
    same order of terms: the results are the same bits -- checked over repeated runs (the two waves a SIMD holds
    run independently of each other), many channel counts (few and many chunks, odd and even), planes smaller than
    a wave's share and larger than a tile, the motion encoder's full shape, scale + bias + ReLU, a channel slice."""
    import os, random
    from visual_odometry_amd import ops
    rnd = random.Random(5)
    shapes = [(2, 64, 60, 80, 64), (1, 128, 120, 160, 256), (2, 112, 120, 160, 256), (3, 336, 60, 80, 256), (2, 256, 60, 80, 576),
              (2, 64, 8, 6, 128), (24, 336, 60, 80, 256), (5, 576, 30, 40, 64), (7, 608, 15, 20, 192), (3, 80, 33, 44, 320),
              (3, 256, 120, 160, 144), (2, 128, 17, 20, 72), (1, 96, 9, 12, 200)]          # last channel block partial
    shapes += [(rnd.randint(1, 5), 16 * rnd.randint(4, 38), rnd.randint(1, 70), 4 * rnd.randint(1, 40), 64 * rnd.randint(1, 9)) for _ in range(14)]

    def template(*args, **kw):
        return ops.conv1x1_split(*args, use_template=True, **kw)

    for (N, C, H, W, K) in shapes:
        x = _rand(N, C, H, W, seed=400 + C).to(DEV)
        w = (_rand(K, C, 1, 1, seed=401) / C ** 0.5).to(DEV)
        # scale and shift as the LAST bytes of an allocation of their own: a partial last channel block (K = 72, 144, 200)
        # must not read past them
        tail_b, tail_s = torch.empty(1 << 18, device=DEV), torch.empty(1 << 18, device=DEV)
        b, sc = tail_b[-K:], tail_s[-K:]
        b.copy_(_rand(K, seed=402)); sc.copy_(_rand(K, seed=403).abs() + 0.5)
        A = ops.conv1x1_split_filter(w)
        rec = {}
        ops.set_recorder(rec)
        got = ops.conv1x1_split(x, A, K)
        ops.set_recorder(None)
        assert "vo_conv1x1_bf16x3" in rec
        want, want2 = template(x, A, K), template(x, A, K, sc, b, True)
        for _ in range(3):
            assert torch.equal(ops.conv1x1_split(x, A, K), want), (N, C, H, W, K)
            assert torch.equal(ops.conv1x1_split(x, A, K, sc, b, True), want2), (N, C, H, W, K)
        assert torch.equal(got, want)
        wide = torch.full((N, K + 8, H, W), 2.5, device=DEV)
        ops.conv1x1_split(x, A, K, None, b, False, out=wide[:, 4:4 + K])
        assert torch.equal(wide[:, 4:4 + K], template(x, A, K, None, b, False))
        assert torch.all(wide[:, :4] == 2.5) and torch.all(wide[:, 4 + K:] == 2.5)


def test_instance_norm_join_normalises_a_raw_skip_branch(vo):
    """vo_instance_norm_apply with skip statistics: relu(act(norm(skip)) + relu(norm(x))) from two raw tensors
    and their plane sums, against the stock operators, with and without the ReLU on the skip branch."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x, s = _rand(2, 5, 13, 18, seed=150).to(DEV) * 3 + 1, _rand(2, 5, 13, 18, seed=151).to(DEV) * 0.5 - 2
    sums = lambda t: torch.stack([t.double().sum((2, 3)), (t.double() ** 2).sum((2, 3))], dim=2).contiguous()
    for skip_relu in (True, False):
        sn = Fn.instance_norm(s, eps=1e-3)
        want = torch.relu((torch.relu(sn) if skip_relu else sn) + torch.relu(Fn.instance_norm(x, eps=1e-5)))
        got = ops.instance_norm_apply_(x.clone(), sums(x), 1e-5, True, s, sums(s), 1e-3, skip_relu)
        assert (got - want).abs().max().item() < 1e-5
    with pytest.raises(ValueError):
        ops.instance_norm_apply_(x.clone(), sums(x), 1e-5, True, None, sums(s))


def test_encoder_with_deferred_norms_matches_stock_operators(vo):
    """BasicEncoder (instance norm) on a one-plane gray input: the stem leaves its sums and is never normalised
    in memory (the first block's convolution and residual join normalise it as they read), the stride-2
    blocks' shortcuts likewise; against the stock-operator forward of the same module on the replicated
    three-channel input."""
    from visual_odometry_amd.raft import layers
    from visual_odometry_amd.raft.layers import BasicEncoder
    torch.manual_seed(5)
    for down in (3, 2):
        enc = BasicEncoder(output_dim=256, norm_fn="instance", downsample=down).to(DEV).eval()
        g = (_rand(2, 1, 192, 256, seed=160) * 0.5).to(DEV)
        with torch.no_grad():
            x3 = g.expand(-1, 3, -1, -1).contiguous()
            want = enc.conv2(enc.layer3(enc.layer2(enc.layer1(torch.relu(enc.norm1(enc.conv1(x3)))))))
            with layers.fused_passes():
                rec = {}
                old = ops_recorder(rec)
                try:
                    got = enc(g)
                finally:
                    ops_recorder(old)
        assert "vo_instance_norm_relu" not in rec                   # no stand-alone statistics pass left
        # the stems on the 7x7 kernel; the stride-2 blocks' 3x3 convolution and shortcut in one pass over their input
        assert "vo_conv7x7_bf16x3" in rec and "vo_conv3x3s2_shortcut_bf16x3" in rec and "vo_conv1x1s2_bf16x3" not in rec
        assert (got - want).abs().max().item() < 2e-4 * max(1.0, want.abs().max().item())


def ops_recorder(rec):
    from visual_odometry_amd import ops
    return ops.set_recorder(rec)


def test_networks_agree_across_conv_arithmetic(vo):
    """Flow and disparity at 480x640 (every 3x3 layer large enough for the split kernels) under the three product
    arithmetics -- float16 pieces (the default), bf16 pieces, float32 matrix instructions only -- against each
    other and against the stock-operator restatement: all inside the 0.05 px budget by three orders of
    magnitude.  The arithmetic is a field of each model: a second model in the same process keeps its own."""
    from visual_odometry_amd.raft.models import RAFTFlow, RAFTStereo, RAFTStereoConfig
    torch.manual_seed(12)
    flow, stereo = RAFTFlow().to(DEV).eval(), RAFTStereo(RAFTStereoConfig.realtime()).to(DEV).eval()
    assert flow.conv_arithmetic == "f16x3" and stereo.conv_arithmetic == "f16x3"
    left, right = _images(1, 480, 640, seed=8)
    out = {}
    for kind in ("f32", "bf16x3", "f16x3"):
        flow.conv_arithmetic = stereo.conv_arithmetic = kind
        rec = {}
        old = ops_recorder(rec)
        try:
            out[kind] = (flow(left[:1], left[1:2], iters=6)[1], stereo(left[:1], right[:1], iters=6)[1])
        finally:
            ops_recorder(old)
        # the kernels that ran are the arithmetic's own
        assert ("vo_conv3x3_winograd4" in rec) == (kind == "f32")
        assert ("vo_conv3x3_bf16x3" in rec) == (kind != "f32")
        # the flow volume: cut rows (bf16 by the encoder's store, float16 by a pass behind the per-image maxima) or float32
        assert ("vo_corr_volume_2d_presplit" in rec) == (kind != "f32")
        assert ("vo_corr_volume_2d" in rec) == (kind == "f32")
        assert ("vo_feature_rows_amax" in rec) == (kind == "f16x3")
        assert ("vo_corr_volume_1d_bf16x3" in rec) == (kind != "f32")
    with pytest.raises(ValueError):
        flow.conv_arithmetic = "fp8"
        flow(left[:1], left[1:2], iters=1)
    flow.conv_arithmetic = "f16x3"
    assert not torch.equal(out["f32"][0], out["bf16x3"][0]) and not torch.equal(out["f16x3"][0], out["bf16x3"][0])
    _, f_ref = raft_ref.raft_flow_forward(flow, left[:1], left[1:2], iters=6)
    _, d_ref = raft_ref.raft_stereo_forward(stereo, left[:1], right[:1], iters=6)
    err = {}
    for kind, (f, d) in out.items():
        err[kind] = (_epe(f, f_ref), (d - d_ref).abs().mean().item())
        assert err[kind][0] < 1e-3 and err[kind][1] < 1e-3, (kind, err[kind])
        assert _epe(f, out["f32"][0]) < 1e-3 and (d - out["f32"][1]).abs().mean().item() < 1e-3
    print("px error against the stock-operator restatement (flow EPE, disparity):", err)


@pytest.mark.parametrize("N,C,H,W,K", [(1, 8, 5, 7, 64), (2, 16, 33, 21, 40), (2, 64, 60, 80, 126), (1, 24, 1, 1, 33),
                                       (2, 384, 30, 40, 256), (1, 8, 70, 3, 32)])
@pytest.mark.parametrize("vertical", [False, True])
def test_winograd_conv5_matches_direct_convolution(vo, N, C, H, W, K, vertical):
    """The one-dimensional F(4,5) Winograd/MFMA kernel (1x5 and 5x1 GRU gate convolutions) against
    the float64 direct convolution: ragged sizes, partial channel blocks, run-to-run identical.
    Tolerance 2e-5 of the largest output (measured 4e-6; MIOpen's float32 result is at 2e-6)."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=110).to(DEV)
    shape = (K, C, 5, 1) if vertical else (K, C, 1, 5)
    w = (_rand(*shape, seed=111) / (5 * C) ** 0.5).to(DEV)
    U = ops.winograd1d_filter(w)
    got = ops.conv5_winograd(x, U, K, vertical)
    want = Fn.conv2d(x.double(), w.double(), None, 1, (2, 0) if vertical else (0, 2))
    assert got.shape == want.shape
    assert (got.double() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
    assert torch.equal(got, ops.conv5_winograd(x, U, K, vertical))
    with pytest.raises(ValueError):
        ops.winograd1d_filter(_rand(8, 6, 1, 5, seed=1).to(DEV))
    with pytest.raises(ValueError):
        ops.conv5_winograd(x, U[:-1].contiguous(), K, vertical)


def test_separable_gru_on_the_winograd_kernel(vo):
    """SepConvGRU (RAFT-Flow's update block) through the fused path, whose 1x5 / 5x1 gate
    convolutions run on the F(4,5) kernel, against the stock operators."""
    from visual_odometry_amd.raft.layers import fused_passes
    from visual_odometry_amd.raft.update import SepConvGRU
    torch.manual_seed(5)
    gru = SepConvGRU(128, 256).to(DEV).eval()
    h, xin = torch.tanh(_rand(2, 128, 30, 40, seed=120)).to(DEV), _rand(2, 256, 30, 40, seed=121).to(DEV)
    with torch.no_grad():
        a = gru(h, xin)
        with fused_passes():
            b = gru(h, xin)
    assert (a - b).abs().max().item() < 2e-5


@pytest.mark.parametrize("N,C,H,W,K", [(1, 8, 5, 7, 64), (2, 13, 33, 21, 40), (2, 36, 30, 41, 18), (3, 324, 15, 20, 256),
                                       (1, 5, 1, 1, 3), (2, 64, 24, 32, 96)])
def test_conv1x1_matches_the_stock_convolution(vo, N, C, H, W, K):
    """The streaming MFMA 1x1 convolution against the float64 convolution: channel counts that are
    no multiple of 8, odd pixel counts, partial channel blocks; scale / shift / ReLU at the store;
    output into a channel slice."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=130).to(DEV)
    w = (_rand(K, C, 1, 1, seed=131) / C ** 0.5).to(DEV)
    scale, shift = _rand(K, seed=132).to(DEV), _rand(K, seed=133).to(DEV)
    Wg = ops.conv1x1_filter(w)
    ref = Fn.conv2d(x.double(), w.double())
    tol = 1e-5 * max(1.0, ref.abs().max().item())
    b = lambda v: v.double()[None, :, None, None]
    assert (ops.conv1x1(x, Wg, K).double() - ref).abs().max().item() < tol
    assert (ops.conv1x1(x, Wg, K, None, shift, True).double() - torch.relu(ref + b(shift))).abs().max().item() < tol
    got = ops.conv1x1(x, Wg, K, scale, shift, False)
    assert (got.double() - (ref * b(scale) + b(shift))).abs().max().item() < 3 * tol
    assert torch.equal(got, ops.conv1x1(x, Wg, K, scale, shift, False))
    wide = torch.full((N, 3 + K + 2, H, W), 7.5, device=DEV)
    ops.conv1x1(x, Wg, K, None, shift, True, out=wide[:, 3:3 + K])
    assert (wide[:, 3:3 + K].double() - torch.relu(ref + b(shift))).abs().max().item() < tol
    assert torch.all(wide[:, :3] == 7.5) and torch.all(wide[:, 3 + K:] == 7.5)
    # the pixel-major output is the same numbers in (N,H,W,K) order, bit for bit
    pm = ops.conv1x1_pixel_major(x, Wg, K, scale, shift, False)
    assert pm.shape == (N, H, W, K) and pm.is_contiguous()
    assert torch.equal(pm, got.permute(0, 2, 3, 1))
    assert torch.equal(ops.conv1x1_pixel_major(x, Wg, K, None, shift, True), ops.conv1x1(x, Wg, K, None, shift, True).permute(0, 2, 3, 1))
    with pytest.raises(ValueError):
        ops.conv1x1(x, Wg[:-1].contiguous(), K)
    with pytest.raises(ValueError):
        ops.conv1x1(x, Wg, K, None, shift[:-1].contiguous())
    with pytest.raises(ValueError):
        ops.conv1x1_pixel_major(x, Wg[:-1].contiguous(), K)


@pytest.mark.parametrize("N,C,H,W,K", [(1, 3, 5, 7, 64), (2, 2, 33, 70, 40), (2, 3, 9, 130, 100), (1, 1, 2, 3, 5),
                                       (2, 3, 64, 96, 64)])
@pytest.mark.parametrize("stride", [1, 2])
def test_conv7x7_matches_the_stock_convolution(vo, N, C, H, W, K, stride):
    """The streaming MFMA 7x7 convolution (stems, motion encoders' flow branch) against the float64
    convolution, stride 1 and 2: images smaller than the filter, odd sizes, widths that are no
    multiple of the 64-pixel row segment, partial channel blocks; scale / shift / ReLU at the store."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=140).to(DEV)
    w = (_rand(K, C, 7, 7, seed=141) / (49 * C) ** 0.5).to(DEV)
    scale, shift = _rand(K, seed=142).to(DEV), _rand(K, seed=143).to(DEV)
    Wg = ops.conv7x7_filter(w)
    ref = Fn.conv2d(x.double(), w.double(), None, stride, 3)
    tol = 1e-5 * max(1.0, ref.abs().max().item())
    b = lambda v: v.double()[None, :, None, None]
    got = ops.conv7x7(x, Wg, K, stride=stride)
    assert got.shape == ref.shape and (got.double() - ref).abs().max().item() < tol
    assert (ops.conv7x7(x, Wg, K, None, shift, True, stride).double() - torch.relu(ref + b(shift))).abs().max().item() < tol
    assert (ops.conv7x7(x, Wg, K, scale, shift, False, stride).double() - (ref * b(scale) + b(shift))).abs().max().item() < 3 * tol
    assert torch.equal(got, ops.conv7x7(x, Wg, K, stride=stride))
    with pytest.raises(ValueError):
        ops.conv7x7_filter(_rand(8, 9, 7, 7, seed=1).to(DEV))
    with pytest.raises(ValueError):
        ops.conv7x7(x, Wg[:-1].contiguous(), K)


@pytest.mark.parametrize("N,C,H,W,K", [(1, 8, 5, 7, 64), (2, 16, 33, 70, 40), (2, 24, 9, 131, 100), (1, 8, 1, 1, 3),
                                       (2, 64, 48, 256, 96), (1, 16, 20, 320, 96), (2, 8, 31, 160, 64)])
def test_conv3x3s2_matches_the_stock_convolution(vo, N, C, H, W, K):
    """The streaming MFMA 3x3 stride-2 convolution (encoders' downsampling layers) against the float64
    convolution: odd and even sizes, a single pixel, partial channel blocks (one k-tile launch), widths
    on the two-rows-of-32 variant (160 and 80 pixels out, an odd number of output rows);
    scale / shift / ReLU at the store."""
    import torch.nn.functional as Fn
    from visual_odometry_amd import ops
    x = _rand(N, C, H, W, seed=150).to(DEV)
    w = (_rand(K, C, 3, 3, seed=151) / (9 * C) ** 0.5).to(DEV)
    scale, shift = _rand(K, seed=152).to(DEV), _rand(K, seed=153).to(DEV)
    Wg = ops.conv3x3s2_filter(w)
    ref = Fn.conv2d(x.double(), w.double(), None, 2, 1)
    tol = 1e-5 * max(1.0, ref.abs().max().item())
    b = lambda v: v.double()[None, :, None, None]
    got = ops.conv3x3s2(x, Wg, K)
    assert got.shape == ref.shape and (got.double() - ref).abs().max().item() < tol
    assert (ops.conv3x3s2(x, Wg, K, None, shift, True).double() - torch.relu(ref + b(shift))).abs().max().item() < tol
    assert (ops.conv3x3s2(x, Wg, K, scale, shift, False).double() - (ref * b(scale) + b(shift))).abs().max().item() < 3 * tol
    assert torch.equal(got, ops.conv3x3s2(x, Wg, K))
    with pytest.raises(ValueError):
        ops.conv3x3s2_filter(_rand(8, 6, 3, 3, seed=1).to(DEV))
    with pytest.raises(ValueError):
        ops.conv3x3s2(x, Wg[:-1].contiguous(), K)
    # the projection shortcut: 1x1, stride 2, on the same kernel with one tap
    w1 = (_rand(K, C, 1, 1, seed=154) / C ** 0.5).to(DEV)
    ref1 = Fn.conv2d(x.double(), w1.double(), None, 2, 0)
    got1 = ops.conv1x1s2(x, ops.conv1x1_filter(w1), K, None, shift, False)
    assert got1.shape == ref1.shape and (got1.double() - (ref1 + b(shift))).abs().max().item() < 1e-5 * max(1.0, ref1.abs().max().item())
