"""GPU parity: each HIP kernel (through the C-ABI) vs a plain torch-CPU fp32 reference of the same op on
bf16-rounded inputs.  Tolerances: outputs are bf16 (8 significant bits) with fp32 accumulation, so the
bound is |err| <= rtol*|ref| + atol with rtol ~ 2^-7 and atol a small fraction of the output range
(accumulated rounding of bf16 intermediates)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16).float()


def close(got, want, rtol=1.2e-2, atol=None):
    want = want.float()
    got = got.float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    if atol is None:
        atol = 8e-3 * float(want.abs().max()) + 1e-6
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    bad = ~(err <= bound)
    assert not bad.any(), f"max err {float(err.max()):.4g} (ref max {float(want.abs().max()):.4g}), {int(bad.sum())} bad"


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def to_nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2)


@pytest.fixture(scope="module")
def ops():
    from audioldm_with_lora_amd import ops as o
    return o


@pytest.mark.parametrize("B,Cin,Cout,H,W,stride,tile", [
    (2, 64, 128, 25, 16, 1, 0), (1, 128, 64, 9, 4, 1, 0), (2, 8, 128, 31, 16, 1, 0), (2, 192, 256, 16, 8, 2, 0),
    (8, 128, 128, 50, 16, 1, 1), (2, 128, 128, 13, 7, 1, 3), (2, 128, 128, 13, 7, 1, 4), (1, 640, 320, 32, 2, 1, 2),
    (1, 128, 8, 20, 16, 1, 0), (4, 128, 256, 70, 16, 1, 9), (2, 64, 128, 129, 8, 1, 9),
])
def test_conv3x3(ops, B, Cin, Cout, H, W, stride, tile):
    g = torch.Generator().manual_seed(0)
    x = bf(torch.randn(B, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    b = torch.randn(Cout, generator=g)
    want = F.conv2d(x, w, b, stride=stride, padding=1)
    pw = ops.pack_conv(w.to(DEV), b.to(DEV))
    y = ops.conv(nhwc(x), pw, stride=(stride, stride), pad=(1, 1), tile=tile)
    close(to_nchw(y), want)


@pytest.mark.parametrize("tile", [7, 8, 15, 16])     # 15 / 16: the wave-specialised forms (8 compute + 4 loader waves)
@pytest.mark.parametrize("B,C1,C2,Cout,H,W,up", [
    (2, 128, 0, 128, 50, 16, False),      # level-0 shape, whole tiles
    (3, 64, 0, 128, 37, 16, False),       # ragged height: the last tile of every image is partial
    (2, 128, 64, 256, 21, 8, False),      # virtual concat, two N tiles, W = 8 (16-pixel MFMA rows span two image rows)
    (1, 192, 0, 40, 10, 8, False),        # Cout below the tile width, 3 channel chunks
    (2, 64, 0, 64, 14, 16, True),         # nearest 2x up-sampling folded into the gather (7x8 -> 14x16)
    (1, 256, 128, 128, 250, 16, False),   # config-2 level-0 up-block conv of one image
    (2, 128, 0, 128, 125, 8, (63, 4)),    # the UNet's 63 x 4 -> 125 x 8 up-sampler: nearest to a size that is not 2x
    (2, 64, 64, 64, 13, 16, (5, 8)),      # 5 -> 13 rows (2.6x), 8 -> 16 columns, two sources
])
def test_conv3x3_halo_tiles(ops, tile, B, C1, C2, Cout, H, W, up):
    """3x3/s1/p1 conv through the LDS-halo kernel == F.conv2d, incl. time-embedding row bias, residual and SiLU epilogue."""
    g = torch.Generator().manual_seed(7)
    ih, iw = up if isinstance(up, tuple) else (H // 2, W // 2) if up else (H, W)
    x1 = bf(torch.randn(B, C1, ih, iw, generator=g))
    x2 = bf(torch.randn(B, C2, ih, iw, generator=g)) if C2 else None
    w = bf(torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt(9 * (C1 + C2)))
    b = torch.randn(Cout, generator=g)
    rb = torch.randn(B, Cout, generator=g)
    r = bf(torch.randn(B, Cout, H, W, generator=g))
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    if up:
        xin = F.interpolate(xin, size=(H, W), mode="nearest")
    want = F.conv2d(xin, w, b, padding=1) + rb[:, :, None, None] + r
    pw = ops.pack_conv(w.to(DEV), b.to(DEV))
    kw = dict(x2=(nhwc(x2) if C2 else None), pad=(1, 1), up_size=((H, W) if up else None), rowbias=rb.to(DEV), rowbias_ld=Cout,
              res=nhwc(r), tile=tile)
    y = ops.conv(nhwc(x1), pw, **kw)
    assert y.shape == (B, H, W, Cout)
    close(to_nchw(y), want)
    base = ops.conv(nhwc(x1), pw, **dict(kw, tile=2))
    assert float((y.float() - base.float()).abs().max()) <= 2e-2 * float(want.abs().max())     # same math as the generic kernel
    for ring in (2, 4):
        assert torch.equal(ops.conv(nhwc(x1), pw, **dict(kw, ring=ring)), y)                         # ring depth never changes results


def test_conv3x3_halo_rejects_unsupported(ops):
    from audioldm_with_lora_amd._lib import AldmError
    g = torch.Generator().manual_seed(8)
    x = nhwc(bf(torch.randn(1, 64, 12, 16, generator=g)))
    pw3 = ops.pack_conv(torch.randn(64, 64, 3, 3, device=DEV) / 24, None)
    pw1 = ops.pack_conv(torch.randn(64, 64, 1, 1, device=DEV) / 8, None)
    with pytest.raises(AldmError):
        ops.conv(x, pw1, tile=7)                                   # not a 3x3
    with pytest.raises(AldmError):
        ops.conv(x, pw3, pad=(1, 1), stride=(2, 2), tile=7)        # strided
    x5 = nhwc(bf(torch.randn(1, 64, 12, 5, generator=g)))
    with pytest.raises(AldmError):
        ops.conv(x5, pw3, pad=(1, 1), tile=8)                      # width 5 does not divide the tile


def test_conv_splitk_matches(ops):
    g = torch.Generator().manual_seed(1)
    x = bf(torch.randn(2, 320, 16, 2, generator=g))
    w = bf(torch.randn(160, 320, 3, 3, generator=g) / math.sqrt(9 * 320))
    b = torch.randn(160, generator=g)
    r = bf(torch.randn(2, 160, 16, 2, generator=g))
    want = F.conv2d(x, w, b, padding=1) + r
    pw = ops.pack_conv(w.to(DEV), b.to(DEV))
    for s in (1, 3, 8):
        y = ops.conv(nhwc(x), pw, pad=(1, 1), res=nhwc(r), splits=s)
        close(to_nchw(y), want)


@pytest.mark.parametrize("B,C,Cout,H,W,groups,splits", [
    (8, 128, 640, 16, 4, 32, 4),      # one quad slot per thread, UNet L3 shape
    (2, 64, 384, 63, 4, 32, 3),       # two slots, ragged strip (L2 shape)
    (1, 64, 128, 40, 16, 8, 5),       # eight slots; 5 splits exercise the tail of the split loop
    (2, 64, 96, 9, 4, 8, 1),          # not split: the ordinary conv -> GroupNorm pair behind the same call
])
def test_conv_groupnorm_over_split_partials(ops, B, C, Cout, H, W, groups, splits):
    """conv(gn=...) == GroupNorm(SiLU) of the convolution (bias + row bias), the reduce of a split-K launch fused into the norm."""
    g = torch.Generator().manual_seed(17)
    x = bf(torch.randn(B, C, H, W, generator=g))
    w = bf(torch.randn(Cout, C, 3, 3, generator=g) / math.sqrt(9 * C))
    b = torch.randn(Cout, generator=g)
    temb = torch.randn(B, Cout + 40, generator=g)
    gamma, beta = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    conv = F.conv2d(x, w, b, padding=1) + temb[:, 40:, None, None]
    want = F.silu(F.group_norm(conv, groups, gamma, beta, 1e-5))
    pw = ops.pack_conv(w.to(DEV), b.to(DEV))
    td = temb.to(DEV)
    y = ops.conv(nhwc(x), pw, pad=(1, 1), rowbias=td[:, 40:], rowbias_ld=Cout + 40, splits=splits,
                 gn=(gamma.to(DEV), beta.to(DEV), groups, 1e-5, ops.ACT_SILU))
    close(to_nchw(y), want, rtol=1.5e-2)
    # and the unfused pair agrees with it to bf16 rounding of the intermediate
    h = ops.conv(nhwc(x), pw, pad=(1, 1), rowbias=td[:, 40:], rowbias_ld=Cout + 40, splits=splits)
    y2 = ops.groupnorm(h, gamma.to(DEV), beta.to(DEV), groups, 1e-5, ops.ACT_SILU)
    close(y.float().cpu(), y2.float().cpu(), rtol=2e-2)
    # conv2 form: shortcut residual joins the sum, the block output is kept next to its (activation-free) norm
    r = bf(torch.randn(B, Cout, H, W, generator=g))
    want_h = F.conv2d(x, w, b, padding=1) + r
    want_n = F.group_norm(want_h, groups, gamma, beta, 1e-6)
    hk, yn = ops.conv(nhwc(x), pw, pad=(1, 1), res=nhwc(r), splits=splits, gn=(gamma.to(DEV), beta.to(DEV), groups, 1e-6, ops.ACT_NONE),
                      gn_keep=True)
    close(to_nchw(hk), want_h)
    close(to_nchw(yn), want_n, rtol=1.5e-2)


@pytest.mark.parametrize("B,C,Cout,C2,H,W,groups,splits", [
    (8, 128, 640, 384, 16, 4, 32, 4),     # up-block norm1 over cat([h, skip]): 32-channel groups, 20 in h and 12 in the skip
    (2, 64, 384, 0, 63, 4, 32, 3),        # down path: no skip
    (2, 64, 128, 128, 25, 16, 32, 2),     # group width 8
])
def test_conv_deferred_reduce_consumed_by_next_groupnorm(ops, B, C, Cout, C2, H, W, groups, splits):
    """conv(defer=...) hands its split-K partial tiles to the NEXT groupnorm(), which also fills the conv's bf16 output"""
    g = torch.Generator().manual_seed(19)
    x = bf(torch.randn(B, C, H, W, generator=g))
    w = bf(torch.randn(Cout, C, 3, 3, generator=g) / math.sqrt(9 * C))
    b = torch.randn(Cout, generator=g)
    r = bf(torch.randn(B, Cout, H, W, generator=g))
    skip = bf(torch.randn(B, C2, H, W, generator=g) * 1.5 + 0.3) if C2 else None
    gamma, beta = torch.randn(Cout + C2, generator=g), torch.randn(Cout + C2, generator=g)
    want_h = F.conv2d(x, w, b, padding=1) + r
    cat = torch.cat([bf(want_h), skip], 1) if C2 else want_h
    want_n = F.silu(F.group_norm(cat, groups, gamma, beta, 1e-5))
    pw = ops.pack_conv(w.to(DEV), b.to(DEV))
    d = ops.conv(nhwc(x), pw, pad=(1, 1), res=nhwc(r), splits=splits, defer=(Cout + C2, groups))
    assert isinstance(d, ops.Deferred) and d.eff > 1
    from audioldm_with_lora_amd._lib import AldmError
    with pytest.raises(AldmError):                           # the workspace is taken until the consumer norm has run
        ops.conv(nhwc(x), pw, pad=(1, 1))
    y = ops.groupnorm(d, gamma.to(DEV), beta.to(DEV), groups, 1e-5, ops.ACT_SILU, x2=(nhwc(skip) if C2 else None))
    close(to_nchw(ops.tensor_of(d)), want_h)
    close(to_nchw(y), want_n, rtol=1.5e-2)
    with pytest.raises(AldmError):                           # consumed: the handle is dead
        ops.groupnorm(d, gamma.to(DEV), beta.to(DEV), groups, 1e-5)
    # a shape the fused norm cannot take (group width does not divide the conv's channels) is not deferred
    t = ops.conv(nhwc(x), pw, pad=(1, 1), splits=splits, defer=(Cout + 16, 4 * groups if (Cout + 16) % (4 * groups) == 0 else groups + 1))
    assert torch.is_tensor(t)


def test_conv_two_sources_rowbias_residual_f32out(ops):
    g = torch.Generator().manual_seed(2)
    x1 = bf(torch.randn(2, 96, 10, 8, generator=g))
    x2 = bf(torch.randn(2, 64, 10, 8, generator=g))
    w = bf(torch.randn(96, 160, 3, 3, generator=g) / math.sqrt(9 * 160))
    b = torch.randn(96, generator=g)
    temb = torch.randn(2, 300, generator=g)
    res = bf(torch.randn(2, 96, 10, 8, generator=g))
    want = F.conv2d(torch.cat([x1, x2], 1), w, b, padding=1) + temb[:, 100:196, None, None] + res
    pw = ops.pack_conv(w.to(DEV), b.to(DEV))
    td = temb.to(DEV)
    y = ops.conv(nhwc(x1), pw, x2=nhwc(x2), pad=(1, 1), rowbias=td[:, 100:], rowbias_ld=300, res=nhwc(res), out_f32=True)
    assert y.dtype == torch.float32
    close(to_nchw(y), want, rtol=2e-3, atol=2e-3 * float(want.abs().max()))


@pytest.mark.parametrize("ih,iw,uh,uw", [(32, 2, 63, 4), (63, 4, 125, 8), (8, 4, 16, 8)])
def test_conv_upsample_fold(ops, ih, iw, uh, uw):
    g = torch.Generator().manual_seed(3)
    x = bf(torch.randn(2, 64, ih, iw, generator=g))
    w = bf(torch.randn(64, 64, 3, 3, generator=g) / 24)
    b = torch.randn(64, generator=g)
    want = F.conv2d(F.interpolate(x, size=(uh, uw), mode="nearest"), w, b, padding=1)
    y = ops.conv(nhwc(x), ops.pack_conv(w.to(DEV), b.to(DEV)), pad=(1, 1), up_size=(uh, uw))
    assert tuple(y.shape) == (2, uh, uw, 64)
    close(to_nchw(y), want)


@pytest.mark.parametrize("M,K,N,r,tile", [(504, 96, 96, 4, 0), (2000, 256, 256, 8, 1), (130, 160, 160, 16, 2), (64, 640, 640, 4, 0)])
def test_linear_lora_fused(ops, M, K, N, r, tile):
    """peft lora.Linear: y = x W^T + b + (alpha/r) (x A^T) B^T, fused in one launch."""
    g = torch.Generator().manual_seed(4)
    x = bf(torch.randn(M, K, generator=g))
    w = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g)
    A = bf(torch.randn(r, K, generator=g) / r)
    Bm = bf(torch.randn(N, r, generator=g) * 0.05)
    s = 2.0
    res = bf(torch.randn(M, N, generator=g))
    want = x @ w.t() + b + s * (x @ A.t()) @ Bm.t() + res
    pw = ops.pack_linear(w.to(DEV), b.to(DEV))
    ops.attach_lora(pw, [(0, N, A.to(DEV), Bm.to(DEV), s)])
    y = ops.linear(x.to(torch.bfloat16).to(DEV), pw, res=res.to(torch.bfloat16).to(DEV), tile=tile)
    close(y, want)
    # LoRA with B = 0 must reproduce the base GEMM bit-for-bit
    pw0 = ops.pack_linear(w.to(DEV), b.to(DEV))
    y0 = ops.linear(x.to(torch.bfloat16).to(DEV), pw0, tile=tile)
    ops.attach_lora(pw0, [(0, N, A.to(DEV), torch.zeros_like(Bm).to(DEV), s)])
    y1 = ops.linear(x.to(torch.bfloat16).to(DEV), pw0, tile=tile)
    assert torch.equal(y0, y1)


def test_qkv_lora_vt_store(ops):
    g = torch.Generator().manual_seed(5)
    B, N, Cc, r = 2, 252, 96, 4
    x = bf(torch.randn(B * N, Cc, generator=g))
    ws = [bf(torch.randn(Cc, Cc, generator=g) / math.sqrt(Cc)) for _ in range(3)]
    As = [bf(torch.randn(r, Cc, generator=g) / r) for _ in range(3)]
    Bs = [bf(torch.randn(Cc, r, generator=g) * 0.05) for _ in range(3)]
    pw = ops.pack_linear(torch.cat(ws).to(DEV), None)
    ops.attach_lora(pw, [(i * Cc, Cc, As[i].to(DEV), Bs[i].to(DEV), 1.0) for i in (0, 2)])   # q and v only
    npad = (N + 7) // 8 * 8
    vt = torch.zeros(B, Cc, npad, dtype=torch.bfloat16, device=DEV)
    qk = ops.conv(x.to(torch.bfloat16).to(DEV).view(B, 1, N, Cc), pw, vt=vt, vt_col0=2 * Cc, vt_ld=npad,
                  vt_batch_stride=Cc * npad).view(B * N, 2 * Cc)
    q = x @ ws[0].t() + (x @ As[0].t()) @ Bs[0].t()
    k = x @ ws[1].t()
    v = x @ ws[2].t() + (x @ As[2].t()) @ Bs[2].t()
    close(qk[:, :Cc], q)
    close(qk[:, Cc:], k)
    close(vt[:, :, :N].permute(0, 2, 1).reshape(B * N, Cc), v)


def test_geglu(ops):
    g = torch.Generator().manual_seed(6)
    M, Cc = 300, 96
    x = bf(torch.randn(M, Cc, generator=g))
    w = bf(torch.randn(8 * Cc, Cc, generator=g) / math.sqrt(Cc))
    b = torch.randn(8 * Cc, generator=g)
    y = x @ w.t() + b
    want = y[:, :4 * Cc] * F.gelu(y[:, 4 * Cc:])
    got = ops.linear(x.to(torch.bfloat16).to(DEV), ops.pack_geglu(w.to(DEV), b.to(DEV)))
    assert got.shape == (M, 4 * Cc)
    close(got, want)
    got2 = ops.linear(x.to(torch.bfloat16).to(DEV), ops.pack_geglu(w.to(DEV), b.to(DEV)), splits=2)
    close(got2, want)


@pytest.mark.parametrize("B,HW,C1,C2,groups,act", [(2, (25, 16), 128, 0, 32, 1), (2, (7, 4), 96, 64, 8, 1), (1, (63, 4), 384, 256, 32, 0), (1, (200, 64), 128, 0, 32, 1)])
def test_groupnorm(ops, B, HW, C1, C2, groups, act):
    g = torch.Generator().manual_seed(7)
    H, W = HW
    x = bf(torch.randn(B, C1 + C2, H, W, generator=g) * 2 + 0.5)
    gm, bt = torch.randn(C1 + C2, generator=g), torch.randn(C1 + C2, generator=g)
    want = F.group_norm(x, groups, gm, bt, eps=1e-5)
    if act:
        want = F.silu(want)
    x1 = nhwc(x[:, :C1])
    x2 = nhwc(x[:, C1:]) if C2 else None
    y = ops.groupnorm(x1, gm.to(DEV), bt.to(DEV), groups, 1e-5, act, x2=x2)
    close(to_nchw(y), want)


@pytest.mark.parametrize("B,H,W,C1,C2,groups,tile", [
    (3, 130, 16, 128, 0, 32, 0),       # 2080 pixels per image: 64 / 128-row M-tiles cross image boundaries (slot 1)
    (2, 128, 16, 128, 128, 32, 7),     # halo tiles (image-aligned) for x; second source from a generic launch; Cg = 8
    (2, 250, 16, 192, 192, 32, 0),     # Cg = 12: a 16-byte unit spans two groups; 48 units per pixel (idle remainder threads)
    (1, 136, 16, 64, 0, 8, 2),
    (2, 250, 16, 128, 0, 32, 6),       # the 8-wave 128x128 tile
    (2, 250, 16, 128, 0, 32, 1),
    (2, 250, 16, 128, 0, 32, 3),
    (2, 250, 16, 128, 0, 32, 4),
    (2, 128, 16, 128, 0, 32, 8),       # halo 64x128
    (2, 250, 16, 256, 128, 32, 0),     # Cg = 12 and C1 % 12 != 0: group 21 takes channels 252..255 from x and 0..7 from x2 (UNet up block 3)
    (2, 250, 16, 384, 256, 32, 0),     # Cg = 20: group 19 straddles the concatenation
    (1, 1024, 16, 128, 0, 32, 2),      # 256 producer tiles per image: the statistics are summed once by the finalize launch
    (2, 512, 16, 128, 128, 32, 0),     # two sources, >= 96 tiles each (finalize with a concatenation)
])
def test_groupnorm_apply_from_conv_statistics(ops, B, H, W, C1, C2, groups, tile):
    """GroupNorm whose statistics were handed over by the producing convolutions (aldm_igemm qstat_out ->
    aldm_groupnorm_apply): conv3x3 -> [cat with a second conv's output] -> GroupNorm(+SiLU) vs torch."""
    g = torch.Generator().manual_seed(31)
    x = bf(torch.randn(B, 64, H, W, generator=g))
    w1 = bf(torch.randn(C1, 64, 3, 3, generator=g) * 0.06)
    b1 = torch.randn(C1, generator=g)
    y1 = ops.conv(nhwc(x), ops.pack_conv(w1.to(DEV), b1.to(DEV)), pad=(1, 1), tile=tile, ring=((2 if tile == 6 else 3) if tile else 0), splits=1,
                  qstats=True)
    assert getattr(y1, "qstats", None) is not None, "the convolution did not leave statistics"
    want1 = F.conv2d(x, w1, b1, padding=1)
    y2 = want2 = None
    if C2:
        w2 = bf(torch.randn(C2, 64, 1, 1, generator=g) * 0.2)
        y2 = ops.conv(nhwc(x), ops.pack_conv(w2.to(DEV), None), splits=1, qstats=True)
        assert getattr(y2, "qstats", None) is not None
        want2 = F.conv2d(x, w2)
    gm, bt = torch.randn(C1 + C2, generator=g), torch.randn(C1 + C2, generator=g)
    before = ops.PROFILE
    ops.PROFILE = []
    got = ops.groupnorm(y1, gm.to(DEV), bt.to(DEV), groups, 1e-5, 1, x2=y2)
    labels = [r[0] for r in ops.PROFILE]
    ops.PROFILE = before
    assert labels and labels[0].startswith("groupnorm_apply"), labels
    # reference on the bf16-rounded conv outputs (what the norm really reads)
    xin = torch.cat([to_nchw(y1)] + ([to_nchw(y2)] if C2 else []), 1)
    close(to_nchw(y1), want1, rtol=2e-2)
    want = F.silu(F.group_norm(xin, groups, gm, bt, 1e-5))
    close(to_nchw(got), want, rtol=2e-2)
    # and the stand-alone kernel on the same tensors agrees
    y1b = y1.clone()
    y2b = y2.clone() if C2 else None
    ref = ops.groupnorm(y1b, gm.to(DEV), bt.to(DEV), groups, 1e-5, 1, x2=y2b)
    close(to_nchw(got), to_nchw(ref).float(), rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("B,H,tile", [(2, 250, 0), (1, 128, 0), (3, 131, 0), (2, 250, 7), (2, 133, 2), (8, 250, 6)])
def test_gn_silu_conv_out_one_launch(ops, B, H, tile):
    """conv_norm_out -> SiLU -> conv_out as ONE launch (aldm_gn_silu_conv3x3_small) against torch and against the two launches it
    replaces: statistics from the producing convolution's table, zero padding applied to the ACTIVATION, every row count."""
    g = torch.Generator().manual_seed(41)
    W, C, Co = 16, 128, 8
    x = bf(torch.randn(B, 64, H, W, generator=g))
    w1 = bf(torch.randn(C, 64, 3, 3, generator=g) * 0.06)
    b1 = torch.randn(C, generator=g)
    y1 = ops.conv(nhwc(x), ops.pack_conv(w1.to(DEV), b1.to(DEV)), pad=(1, 1), tile=tile, ring=(3 if tile else 0), splits=1, qstats=True)
    assert getattr(y1, "qstats", None) is not None
    gm, bt = torch.randn(C, generator=g) * 0.3 + 1, torch.randn(C, generator=g) * 0.2
    w2 = bf(torch.randn(Co, C, 3, 3, generator=g) * 0.05)
    b2 = torch.randn(Co, generator=g)
    pw = ops.pack_conv(w2.to(DEV), b2.to(DEV))
    assert ops.gn_silu_conv_out_ok(y1, pw, 32)
    got = ops.gn_silu_conv_out(y1, gm.to(DEV), bt.to(DEV), 32, 1e-5, pw)
    assert got.dtype == torch.float32 and tuple(got.shape) == (B, H, W, Co)
    a = bf(F.silu(F.group_norm(to_nchw(y1).float(), 32, gm, bt, 1e-5)))               # the activation as the kernel rounds it (bf16 in LDS)
    want = F.conv2d(a, w2, b2, padding=1)
    close(got.permute(0, 3, 1, 2), want, rtol=2e-2)
    two = ops.conv(ops.groupnorm(y1, gm.to(DEV), bt.to(DEV), 32, 1e-5, 1), pw, pad=(1, 1), out_f32=True)
    close(got.permute(0, 3, 1, 2), two.permute(0, 3, 1, 2).float().cpu(), rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("B,H,C1,C2,N,res", [(2, 250, 128, 0, 128, True), (2, 250, 256, 128, 128, False), (1, 130, 128, 128, 128, False),
                                               (2, 250, 128, 0, 256, False)])
@pytest.mark.parametrize("tile", [0, 15, 16])    # 0: the default halo tile; 15 / 16: the loader waves normalise the halo
def test_conv_applies_groupnorm_of_its_input(ops, B, H, C1, C2, N, res, tile):
    """ops.conv(gn_in=): ResnetBlock2D norm1 / norm2 + SiLU inside the consuming convolution's halo tile (statistics from the
    producers' tables, concatenated sources, groups that straddle the concatenation) against torch and against the two launches."""
    g = torch.Generator().manual_seed(43)
    W = 16
    x0 = bf(torch.randn(B, 64, H, W, generator=g))
    def producer(co):
        w = bf(torch.randn(co, 64, 3, 3, generator=g) * 0.06)
        b = torch.randn(co, generator=g)
        y = ops.conv(nhwc(x0), ops.pack_conv(w.to(DEV), b.to(DEV)), pad=(1, 1), splits=1, qstats=True)
        assert getattr(y, "qstats", None) is not None
        return y
    y1 = producer(C1)
    y2 = producer(C2) if C2 else None
    C = C1 + C2
    gm, bt = torch.randn(C, generator=g) * 0.3 + 1, torch.randn(C, generator=g) * 0.2
    w = bf(torch.randn(N, C, 3, 3, generator=g) * 0.03)
    b = torch.randn(N, generator=g)
    pw = ops.pack_conv(w.to(DEV), b.to(DEV))
    r = bf(torch.randn(B, N, H, W, generator=g)) if res else None
    rb = torch.randn(B, N, generator=g)
    assert ops.gn_in_ok(y1, y2, pw)
    kw = dict(pad=(1, 1), rowbias=rb.to(DEV), rowbias_ld=N, res=(nhwc(r) if res else None))
    got = ops.conv(y1, pw, x2=y2, gn_in=(gm.to(DEV), bt.to(DEV), 32, 1e-5, 1), qstats=True, tile=tile, **kw)
    assert getattr(got, "qstats", None) is not None                       # and it hands ITS statistics on
    xin = torch.cat([to_nchw(y1)] + ([to_nchw(y2)] if C2 else []), 1)
    a = bf(F.silu(F.group_norm(xin, 32, gm, bt, 1e-5)))
    want = F.conv2d(a, w, b, padding=1) + rb[:, :, None, None] + (r if res else 0)
    close(to_nchw(got), want, rtol=2e-2)
    two = ops.conv(ops.groupnorm(y1, gm.to(DEV), bt.to(DEV), 32, 1e-5, 1, x2=y2), pw, **kw)
    close(to_nchw(got), to_nchw(two), rtol=1e-2, atol=3e-2)
    nxt = ops.groupnorm(got, torch.ones(N).to(DEV), torch.zeros(N).to(DEV), 32, 1e-5, 0)      # consumer of the handed-over statistics
    close(to_nchw(nxt), F.group_norm(to_nchw(got), 32, None, None, 1e-5), rtol=2e-2)


def test_layernorm(ops):
    g = torch.Generator().manual_seed(8)
    for Cc in (64, 96, 256, 640):
        x = bf(torch.randn(77, Cc, generator=g) * 3 + 1)
        gm, bt = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
        y = ops.layernorm(x.to(torch.bfloat16).to(DEV), gm.to(DEV), bt.to(DEV))
        close(y, F.layer_norm(x, (Cc,), gm, bt, 1e-5))


@pytest.mark.parametrize("B,N,H,d", [(2, 1000, 8, 32), (2, 252, 8, 48), (2, 64, 8, 80), (1, 1008, 4, 16), (2, 256, 4, 24), (1, 64, 4, 40), (1, 16, 4, 40), (1, 1024, 8, 32)])
def test_attention(ops, B, N, H, d):
    g = torch.Generator().manual_seed(9)
    Cc = H * d
    q, k, v = (bf(torch.randn(B, N, Cc, generator=g)) for _ in range(3))
    q = bf(q * 1.5)
    sp = lambda z: z.view(B, N, H, d).transpose(1, 2)
    want = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B * N, Cc)
    qk = torch.cat([q, k], -1).view(B * N, 2 * Cc).to(torch.bfloat16).to(DEV)
    npad = (N + 7) // 8 * 8
    vt = torch.full((B, Cc, npad), float("nan"), dtype=torch.bfloat16, device=DEV)   # padding must be ignored
    vt[:, :, :N] = v.transpose(1, 2).to(torch.bfloat16).to(DEV)
    out = ops.attention(qk, vt, B, N, H, d)
    close(out, want, rtol=2e-2, atol=1e-2)


def test_attention_online_softmax_rescale_branch(ops):
    """Force the running max to jump at a late key tile (guide rule 26): spike one key against one query."""
    g = torch.Generator().manual_seed(10)
    B, N, H, d = 1, 320, 8, 32
    Cc = H * d
    q, k, v = (bf(torch.randn(B, N, Cc, generator=g)) for _ in range(3))
    k[0, 300] = bf(q[0, 5] * 4.0)          # key 300 (5th tile) dominates query 5's row
    sp = lambda z: z.view(B, N, H, d).transpose(1, 2)
    want = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B * N, Cc)
    qk = torch.cat([q, k], -1).view(B * N, 2 * Cc).to(torch.bfloat16).to(DEV)
    vt = v.transpose(1, 2).contiguous().to(torch.bfloat16).to(DEV)
    close(ops.attention(qk, vt, B, N, H, d), want, rtol=2e-2, atol=1e-2)


@pytest.mark.parametrize("prescaled", [False, True])
def test_attention_prescaled_path_and_deferred_rescale(ops, prescaled):
    """The UNet's path hands the kernel a Q that already carries d^-0.5 log2(e); a spike at a late key tile (more than
    RESCALE_THR above the running max) must take the deferred-rescale branch, small increases must not change the result."""
    g = torch.Generator().manual_seed(12)
    B, N, H, d = 2, 1000, 8, 32
    Cc = H * d
    q, k, v = (bf(torch.randn(B, N, Cc, generator=g)) for _ in range(3))
    k[0, 900] = bf(q[0, 7] * 6.0)
    k[1, 130] = bf(q[1, 999] * 3.0)
    sp = lambda z: z.view(B, N, H, d).transpose(1, 2)
    want = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B * N, Cc)
    qs = (q * (ops.LOG2E / math.sqrt(d))) if prescaled else q
    qk = torch.cat([qs, k], -1).view(B * N, 2 * Cc).to(torch.bfloat16).to(DEV)
    vt = v.transpose(1, 2).contiguous().to(torch.bfloat16).to(DEV)
    close(ops.attention(qk, vt, B, N, H, d, prescaled=prescaled), want, rtol=3e-2, atol=1.5e-2)


@pytest.mark.parametrize("B,N,d", [(2, 200, 128), (1, 1000, 256), (2, 2048, 512), (1, 4000, 512), (1, 37, 512)])
def test_attention_wide_head(ops, B, N, d):
    """AutoencoderKL mid-block attention: 1 head, d = C, flash-style on 16x16x32 MFMA tiles (aldm_attention_wide), ragged N,
    keys zero-padded to a multiple of 32, plus a forced late rescale."""
    g = torch.Generator().manual_seed(13)
    q, k, v = (bf(torch.randn(B, N, d, generator=g)) for _ in range(3))
    q = bf(q * 2.0)
    if N > 100:
        k[0, N - 10] = bf(q[0, 3] * 1.5)                 # a key far down the sequence dominates query 3
    want = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0].reshape(B * N, d)
    qs = q * (ops.LOG2E / math.sqrt(d))
    qk = torch.cat([qs, k], -1).view(B * N, 2 * d).to(torch.bfloat16).to(DEV)
    npad = (N + 31) // 32 * 32
    vt = torch.zeros(B, d, npad, dtype=torch.bfloat16, device=DEV)
    vt[:, :, :N] = v.transpose(1, 2).to(torch.bfloat16).to(DEV)
    out = ops.attention_wide(qk, vt, B, N, d)
    close(out, want, rtol=3e-2, atol=1.5e-2)


@pytest.mark.parametrize("B,N,r", [(3, 64, 4), (2, 64, 0), (2, 40, 8), (1, 64, 10)])
def test_attn_block64_projection_and_attention_in_one_launch(ops, B, N, r):
    """aldm_attn_block64: LayerNorm (folded, statistics handed over) -> to_q | to_k | to_v with LoRA -> softmax(Q K^T) V per
    (sample, head), against torch fp32 on the bf16-rounded operands.  r = LoRA rank per projection (3 r <= 16: one rank tile,
    <= 32: two); N = 40: ragged tokens (rows beyond N read as zeros, keys masked)."""
    H, d = 8, 80
    Cc = H * d
    g = torch.Generator().manual_seed(31)
    x = bf(torch.randn(B * N, Cc, generator=g) * 1.3 + 0.2)
    wq, wk, wv = (bf(torch.randn(Cc, Cc, generator=g) / math.sqrt(Cc)) for _ in range(3))
    gm, bt = torch.randn(Cc, generator=g) * 0.3 + 1, torch.randn(Cc, generator=g) * 0.2
    xn = F.layer_norm(x, (Cc,), gm, bt, 1e-5)
    lor = [(bf(torch.randn(r, Cc, generator=g) / math.sqrt(Cc)), bf(torch.randn(Cc, r, generator=g) * 0.3), 2.0) if r else None
           for _ in range(3)]
    q, k, v = (xn @ w.t() + (l[2] * (xn @ l[0].t()) @ l[1].t() if l else 0.0) for w, l in zip((wq, wk, wv), lor))
    sp = lambda t: t.view(B, N, H, d).transpose(1, 2)
    want = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B * N, Cc)
    qs = ops.LOG2E / math.sqrt(d)
    pw = ops.pack_linear_ln(torch.cat([wq * qs, wk, wv]).to(DEV), None, gm.to(DEV), bt.to(DEV))
    ops.attach_lora(pw, [None if l is None else (i * Cc, Cc, l[0].to(DEV), l[1].to(DEV), l[2] * (qs if i == 0 else 1.0))
                         for i, l in enumerate(lor)])
    xd = x.to(torch.bfloat16).to(DEV)
    # row partials as a producer epilogue leaves them: [M][np][2] = (sum, sum of squares) per 64-column slab
    xs = xd.float().view(B * N, Cc // 64, 64)
    parts = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).contiguous()
    assert ops.attn_block64_ok(pw, N, H, d, parts)
    out = ops.attn_block64(xd, pw, parts, B, N, H, d)
    # first against the two-launch path it replaces (same packed operands, same bf16 rounding points) ...
    npad = (N + 7) // 8 * 8
    vt = torch.zeros(B, Cc, npad, dtype=torch.bfloat16, device=DEV)
    qk = ops.conv(xd.view(B, 1, N, Cc), pw, vt=vt, vt_col0=2 * Cc, vt_ld=npad, vt_batch_stride=Cc * npad, ln_parts=parts)
    ref = ops.attention(qk.view(B * N, 2 * Cc), vt, B, N, H, d, prescaled=True)
    close(out, ref.float().cpu(), rtol=2e-2)
    # ... then against torch fp32 (no bf16 rounding of q | k | v | p anywhere: the bound is that of the two-launch path)
    close(out, want, rtol=3e-2, atol=2.5e-2 * float(want.abs().max()))
    close(ref, want, rtol=3e-2, atol=2.5e-2 * float(want.abs().max()))
    # BASELINE config 5: the same launch with e4m3 Q / K / V / P operands against the two-launch fp8 path (same quantisation points:
    # the bf16-rounded projection outputs, P scaled by 2^8: rel. L2 <= 2e-2); against torch fp32 its error is that path's (these
    # synthetic scores are peaky -- LoRA-amplified q / k -- and e4m3 quantisation of Q / K moves them: 0.07 - 0.09 for both forms)
    out8 = ops.attn_block64(xd, pw, parts, B, N, H, d, fp8=True)
    ref8 = ops.attention(qk.view(B * N, 2 * Cc), vt, B, N, H, d, prescaled=True, fp8=True)
    rl2 = lambda a, b: float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())
    assert rl2(out8, ref8) < 2e-2, rl2(out8, ref8)
    assert rl2(out8, want) < rl2(ref8, want) + 1e-2, (rl2(out8, want), rl2(ref8, want))
    assert not torch.equal(out8, out)


@pytest.mark.parametrize("B,N,r", [(8, 252, 4), (2, 256, 0), (3, 100, 8), (1, 17, 10), (8, 256, 4)])
def test_attn_block256_projection_and_attention_in_one_launch(ops, monkeypatch, B, N, r):
    """aldm_attn_block256 (the 252-token level: C = 384 = 8 heads x 48): LayerNorm (folded, statistics handed over) -> to_q | to_k | to_v
    with LoRA -> softmax(Q K^T) V per (sample, head), against the two-launch path it replaces and against torch fp32 on the
    bf16-rounded operands.  N = 252 / 256: the inference / training token counts; 100, 17: ragged (whole query tiles and waves of
    padding, keys masked); r = LoRA rank per projection (3 r <= 16: one rank tile, <= 32: two)."""
    H, d = 8, 48
    Cc = H * d
    g = torch.Generator().manual_seed(32)
    x = bf(torch.randn(B * N, Cc, generator=g) * 1.3 + 0.2)
    wq, wk, wv = (bf(torch.randn(Cc, Cc, generator=g) / math.sqrt(Cc)) for _ in range(3))
    gm, bt = torch.randn(Cc, generator=g) * 0.3 + 1, torch.randn(Cc, generator=g) * 0.2
    xn = F.layer_norm(x, (Cc,), gm, bt, 1e-5)
    lor = [(bf(torch.randn(r, Cc, generator=g) / math.sqrt(Cc)), bf(torch.randn(Cc, r, generator=g) * 0.3), 2.0) if r else None
           for _ in range(3)]
    q, k, v = (xn @ w.t() + (l[2] * (xn @ l[0].t()) @ l[1].t() if l else 0.0) for w, l in zip((wq, wk, wv), lor))
    sp = lambda t: t.view(B, N, H, d).transpose(1, 2)
    want = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B * N, Cc)
    qs = ops.LOG2E / math.sqrt(d)
    pw = ops.pack_linear_ln(torch.cat([wq * qs, wk, wv]).to(DEV), None, gm.to(DEV), bt.to(DEV))
    ops.attach_lora(pw, [None if l is None else (i * Cc, Cc, l[0].to(DEV), l[1].to(DEV), l[2] * (qs if i == 0 else 1.0))
                         for i, l in enumerate(lor)])
    xd = x.to(torch.bfloat16).to(DEV)
    xs = xd.float().view(B * N, Cc // 64, 64)
    parts = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).contiguous()
    monkeypatch.setattr(ops, "ATTN_BLOCK256", True)                      # (off by default: slower than the two launches it fuses, ops.py)
    assert ops.attn_block_ok(pw, N, H, d, parts) == 256
    out = ops.attn_block(xd, pw, parts, B, N, H, d)
    npad = (N + 7) // 8 * 8
    vt = torch.zeros(B, Cc, npad, dtype=torch.bfloat16, device=DEV)
    qk = ops.conv(xd.view(B, 1, N, Cc), pw, vt=vt, vt_col0=2 * Cc, vt_ld=npad, vt_batch_stride=Cc * npad, ln_parts=parts)
    ref = ops.attention(qk.view(B * N, 2 * Cc), vt, B, N, H, d, prescaled=True)
    close(out, ref.float().cpu(), rtol=2e-2)
    close(out, want, rtol=3e-2, atol=2.5e-2 * float(want.abs().max()))


@pytest.mark.parametrize("B,T,C,K,d,mrf", [(2, 1000, 32, 3, 1, False), (1, 777, 32, 11, 5, True), (2, 600, 64, 7, 3, True), (1, 300, 64, 11, 5, False),
                                            (1, 50, 32, 7, 1, True), (3, 256, 64, 3, 3, False)])
def test_hifigan_residual_pair_one_launch(ops, B, T, C, K, d, mrf):
    """aldm_hifigan_respair: x + conv2(lrelu(conv1(lrelu(x)))) (one step of HifiGanResidualBlock.forward), optionally folded into the MRF
    mean and the next stage's leaky-relu, against torch fp32 on the bf16-rounded operands and against the two aldm_igemm launches it
    replaces.  Ragged lengths (T % 256 != 0, T < one tile), both channel counts, every tap count, dilations 1 / 3 / 5."""
    g = torch.Generator().manual_seed(77 + K)
    x = bf(torch.randn(B, C, T, generator=g))
    w1, w2 = bf(torch.randn(C, C, K, generator=g) / math.sqrt(C * K)), bf(torch.randn(C, C, K, generator=g) / math.sqrt(C * K))
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    acc = bf(torch.randn(B, C, T, generator=g)) if mrf else None
    t = F.leaky_relu(F.conv1d(F.leaky_relu(x, 0.1), w1, b1, dilation=d, padding=(K * d - d) // 2), 0.1)
    want = x + F.conv1d(bf(t), w2, b2, padding=(K - 1) // 2)
    if mrf:
        want = F.leaky_relu(want / 3 + acc, 0.01)
    cl = lambda v: v.transpose(1, 2).unsqueeze(1).contiguous().to(torch.bfloat16).to(DEV)       # [B, 1, T, C] channels-last
    c1 = ops.pack_conv(w1.unsqueeze(2).to(DEV), b1.to(DEV))
    c2 = ops.pack_conv(w2.unsqueeze(2).to(DEV), b2.to(DEV))
    xd = cl(x)
    assert ops.hifigan_respair_ok(xd, c1, c2, d)
    kw = dict(alpha=1.0 / 3, res2=cl(acc), post_act=ops.ACT_LRELU, post_slope=0.01) if mrf else {}
    got = ops.hifigan_respair(xd, c1, c2, d, 0.1, **kw)
    close(got[:, 0].transpose(1, 2), want, rtol=2e-2)
    # the two launches it replaces (same packed weights, same bf16 rounding of the intermediate)
    ra = ops.conv(xd, ops.pack_conv(torch.eye(C).view(C, C, 1, 1).to(DEV), None), out_act=ops.ACT_LRELU, out_slope=0.1)      # lrelu(x)
    tt = ops.conv(ra, c1, pad=(0, (K * d - d) // 2), dil=(1, d), out_act=ops.ACT_LRELU, out_slope=0.1)
    two = ops.conv(tt, c2, pad=(0, (K - 1) // 2), res=xd, **(dict(alpha=1.0 / 3, res2=cl(acc), post_act=ops.ACT_LRELU, post_slope=0.01) if mrf else {}))
    close(got, two.float().cpu(), rtol=1.5e-2)


@pytest.mark.parametrize("B,T,C,K", [(2, 1000, 32, 7), (1, 130, 64, 3), (3, 257, 8, 11), (1, 5, 32, 7)])
def test_conv1d_to_one_channel_with_tanh(ops, B, T, C, K):
    """aldm_conv1d_to1 (SpeechT5HifiGan's conv_post + tanh) against torch fp32 on the bf16-rounded operands."""
    g = torch.Generator().manual_seed(90 + K)
    x = bf(torch.randn(B, C, T, generator=g))
    w = bf(torch.randn(1, C, K, generator=g) / math.sqrt(C * K))
    b = torch.randn(1, generator=g) * 0.1
    want = torch.tanh(F.conv1d(x, w, b, padding=(K - 1) // 2))[:, 0]
    pw = ops.pack_conv(torch.cat([w, torch.zeros(7, C, K)]).unsqueeze(2).to(DEV), torch.cat([b, torch.zeros(7)]).to(DEV))
    got = ops.conv1d_to1(x.transpose(1, 2).unsqueeze(1).contiguous().to(torch.bfloat16).to(DEV), pw, act=ops.ACT_TANH)
    close(got, want, rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("M_hw,Cin,Cout,splits,tile", [((32, 2), 640, 640, 6, 4), ((63, 4), 384, 384, 1, 2), ((125, 8), 256, 256, 2, 3), ((32, 2), 128, 1280, 1, 2)])
def test_xcd_map_changes_speed_only(ops, monkeypatch, M_hw, Cin, Cout, splits, tile):
    """aldm_igemm_t.xcd_map: the work-item -> XCD order (activation- or weight-stationary, auto) is a permutation of which workgroup
    computes which (M-tile, N-tile, K-split): every order gives bit-identical results, for split-K (partials summed in split order by
    the reduce) and unsplit launches, at the UNet's low-resolution shapes (8 images)."""
    g = torch.Generator().manual_seed(5 + Cin)
    H, W = M_hw
    x = nhwc(bf(torch.randn(8, Cin, H, W, generator=g)))
    pw = ops.pack_conv((torch.randn(Cout, Cin, 3, 3, generator=g) / 40).to(DEV), torch.randn(Cout, generator=g).to(DEV))
    outs = []
    for mode in (0, 1, 2):
        monkeypatch.setattr(ops, "XCD_MAP", mode)
        outs.append(ops.conv(x, pw, pad=(1, 1), splits=splits, tile=tile).clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    want = F.conv2d(to_nchw(x), pw.w[:, :9 * Cin].view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).float().cpu(), pw.bias.cpu(), padding=1)
    close(to_nchw(outs[0]), want, rtol=2e-2)


@pytest.mark.parametrize("tile,splits,ring", [(10, 1, 3), (10, 6, 3), (11, 1, 2), (11, 4, 4)])
def test_conv3x3_eight_wave_small_tiles(ops, tile, splits, ring):
    """the 8-wave forms of the 64x128 / 128x64 tiles (two waves per SIMD at ~one workgroup per CU): same results as torch, split-K or not"""
    g = torch.Generator().manual_seed(60 + tile)
    x, x2 = bf(torch.randn(8, 128, 32, 2, generator=g)), bf(torch.randn(8, 64, 32, 2, generator=g))
    w, b = bf(torch.randn(320, 192, 3, 3, generator=g) * 0.03), torch.randn(320, generator=g)
    r = bf(torch.randn(8, 320, 32, 2, generator=g))
    want = F.conv2d(torch.cat([x, x2], 1), w, b, padding=1) + r
    y = ops.conv(nhwc(x), ops.pack_conv(w.to(DEV), b.to(DEV)), x2=nhwc(x2), pad=(1, 1), res=nhwc(r), tile=tile, ring=ring, splits=splits)
    close(to_nchw(y), want)


@pytest.mark.parametrize("tile,shape,splits", [(7, (8, 32, 2), 10), (7, (8, 32, 2), 3), (7, (3, 63, 4), 4), (8, (3, 63, 4), 5), (7, (2, 125, 8), 2),
                                               (8, (2, 125, 8), 2), (7, (2, 37, 16), 5),
                                               (15, (8, 32, 2), 10), (15, (3, 63, 4), 4), (16, (3, 63, 4), 5), (15, (2, 125, 8), 2), (16, (2, 125, 8), 3),
                                               (15, (2, 37, 16), 5)])
def test_conv3x3_halo_split_k(ops, tile, shape, splits):
    """split-K on the halo tiles (by 64-channel chunk, partial tiles in the workspace slabs): == torch on the latent geometries of every UNet
    level (2 / 4 / 8 / 16 wide; image heights that do not fill the last tile), two sources + residual, more splits asked than chunks
    divide into, and the reduce deferred to the GroupNorm that follows."""
    B, H, W = shape
    g = torch.Generator().manual_seed(140 + tile + splits)
    x, x2 = bf(torch.randn(B, 192, H, W, generator=g)), bf(torch.randn(B, 128, H, W, generator=g))
    w, b = bf(torch.randn(256, 320, 3, 3, generator=g) * 0.03), torch.randn(256, generator=g)
    r = bf(torch.randn(B, 256, H, W, generator=g))
    want = F.conv2d(torch.cat([x, x2], 1), w, b, padding=1) + r
    pw = ops.pack_conv(w.to(DEV), b.to(DEV))
    y = ops.conv(nhwc(x), pw, x2=nhwc(x2), pad=(1, 1), res=nhwc(r), tile=tile, splits=splits)
    close(to_nchw(y), want)
    gamma, beta = torch.randn(256, generator=g).to(DEV), torch.randn(256, generator=g).to(DEV)
    yn = ops.conv(nhwc(x), pw, x2=nhwc(x2), pad=(1, 1), res=nhwc(r), tile=tile, splits=splits, gn=(gamma, beta, 32, 1e-5, ops.ACT_SILU))
    close(to_nchw(yn), F.silu(F.group_norm(want, 32, gamma.cpu(), beta.cpu(), 1e-5)), rtol=3e-2)


@pytest.mark.parametrize("tile", [15, 16])
@pytest.mark.parametrize("K,dil,T,C,N", [(3, 1, 1500, 128, 128), (7, 3, 1000, 128, 256), (11, 5, 777, 256, 128), (11, 1, 2000, 64, 64), (7, 5, 130, 128, 128)])
def test_conv1d_dilated_on_halo_ws_tiles(ops, tile, K, dil, T, C, N):
    """the vocoder's dilated 1-D convolutions on the wave-specialised halo tiles (read as a K x 1 filter over a T x 1 image: the tile's
    BM time steps + (K - 1) dil halo rows are fetched once per 64-channel chunk instead of once per tap): == torch for the HiFi-GAN
    residual-block geometries (K 3 / 7 / 11, dilation 1 / 3 / 5), T that does not fill the last tile, a clip boundary inside the
    batch, the residual + second (leaky-ReLU'd) output epilogue of the block, and split-K."""
    if tile == 16 and 64 + (K - 1) * dil > 128:
        pytest.skip("halo of this filter exceeds the 64-row tile's two DMA passes")
    g = torch.Generator().manual_seed(160 + K + dil)
    B = 3
    x = bf(torch.randn(B, C, 1, T, generator=g))
    w, b = bf(torch.randn(N, C, 1, K, generator=g) / math.sqrt(K * C)), torch.randn(N, generator=g)
    pad = (K - 1) * dil // 2
    want = F.conv2d(x, w, b, padding=(0, pad), dilation=(1, dil))
    pw = ops.pack_conv(w.to(DEV), b.to(DEV))
    y = ops.conv(nhwc(x), pw, pad=(0, pad), dil=(1, dil), tile=tile)
    close(to_nchw(y), want)
    r = bf(torch.randn(B, N, 1, T, generator=g))
    out2 = torch.empty(B, 1, T, N, dtype=torch.bfloat16, device=DEV)
    y = ops.conv(nhwc(x), pw, pad=(0, pad), dil=(1, dil), tile=tile, res=nhwc(r), out2=out2, post_act=ops.ACT_LRELU, post_slope=0.1)
    close(to_nchw(y), want + r)
    close(to_nchw(out2), F.leaky_relu(bf(want + r), 0.1), rtol=2e-2)
    if C >= 128:
        close(to_nchw(ops.conv(nhwc(x), pw, pad=(0, pad), dil=(1, dil), tile=tile, splits=2)), want)
    base = ops.conv(nhwc(x), pw, pad=(0, pad), dil=(1, dil), tile=2)
    assert float((ops.conv(nhwc(x), pw, pad=(0, pad), dil=(1, dil), tile=tile).float() - base.float()).abs().max()) <= 2e-2 * float(want.abs().max())


@pytest.mark.parametrize("tile,shape", [(15, (2, 125, 8)), (15, (2, 37, 16)), (16, (2, 125, 8)), (16, (3, 63, 4)), (16, (2, 37, 16))])
@pytest.mark.parametrize("splits", [1, 2, 3])
def test_conv3x3_halo_ws_fused_shortcut(ops, tile, shape, splits):
    """conv2 + conv_shortcut of a ResnetBlock2D as ONE launch on the wave-specialised halo tiles (the 1x1 segment over x3 | x4 walks its
    chunks as single-tap items with their own halo image, three halo buffers): == torch, whole and split-K (a split takes main chunks
    AND a share of the segment's chunks; 3 splits of 2 main chunks leave the third without a main chunk only if the rule were wrong),
    one or two segment sources, with the reduce deferred to a GroupNorm."""
    B, H, W = shape
    g = torch.Generator().manual_seed(150 + tile + splits)
    h = bf(torch.randn(B, 192, H, W, generator=g))
    xa, xb = bf(torch.randn(B, 128, H, W, generator=g)), bf(torch.randn(B, 64, H, W, generator=g))
    w, b = bf(torch.randn(256, 192, 3, 3, generator=g) * 0.03), torch.randn(256, generator=g)
    for two in (True, False):
        cs = 192 if two else 128
        ws_, bs_ = bf(torch.randn(256, cs, 1, 1, generator=g) * 0.06), torch.randn(256, generator=g)
        xs = torch.cat([xa, xb], 1) if two else xa
        want = F.conv2d(h, w, b, padding=1) + F.conv2d(xs, ws_, bs_)
        pw = ops.pack_conv_shortcut(w.to(DEV), b.to(DEV), ws_.to(DEV), bs_.to(DEV))
        y = ops.conv(nhwc(h), pw, pad=(1, 1), x3=nhwc(xa), x4=(nhwc(xb) if two else None), tile=tile, ring=3, splits=splits)
        close(to_nchw(y), want)
    gamma, beta = torch.randn(256, generator=g).to(DEV), torch.randn(256, generator=g).to(DEV)
    yn = ops.conv(nhwc(h), pw, pad=(1, 1), x3=nhwc(xa), tile=tile, ring=3, splits=splits, gn=(gamma, beta, 32, 1e-5, ops.ACT_SILU))
    close(to_nchw(yn), F.silu(F.group_norm(want, 32, gamma.cpu(), beta.cpu(), 1e-5)), rtol=3e-2)


@pytest.mark.parametrize("tile,ring", [(13, 3), (13, 4), (14, 3), (14, 4), (13, 2), (14, 2)])   # ring 2 = register-staged loaders
@pytest.mark.parametrize("kind", ["splitk_res", "plain_ragged", "shortcut_two_src", "shortcut_splitk", "one_by_one"])
def test_conv_wave_specialised_small_tiles(ops, tile, ring, kind):
    """tiles 13 / 14 (csrc/igemm_ws.hip on 64x128 / 128x64: 4 compute + 4 loader waves) == torch for the launch forms of the UNet's
    low-resolution levels: split-K with residual, ragged M and N, conv2 + conv_shortcut as one GEMM (the fused 1x1 segment over one or
    two block inputs, whole or split-K with the split boundary inside / at the segment), 1x1 over two sources."""
    g = torch.Generator().manual_seed(130 + tile)
    if kind == "splitk_res":
        x = bf(torch.randn(8, 320, 32, 2, generator=g))
        w, b = bf(torch.randn(320, 320, 3, 3, generator=g) * 0.03), torch.randn(320, generator=g)
        r = bf(torch.randn(8, 320, 32, 2, generator=g))
        y = ops.conv(nhwc(x), ops.pack_conv(w.to(DEV), b.to(DEV)), pad=(1, 1), res=nhwc(r), tile=tile, ring=ring, splits=6)
        close(to_nchw(y), F.conv2d(x, w, b, padding=1) + r)
    elif kind == "plain_ragged":
        x = bf(torch.randn(3, 128, 63, 4, generator=g))                   # M = 756: not a multiple of either tile height
        w, b = bf(torch.randn(200, 128, 3, 3, generator=g) * 0.04), torch.randn(200, generator=g)
        close(to_nchw(ops.conv(nhwc(x), ops.pack_conv(w.to(DEV), b.to(DEV)), pad=(1, 1), tile=tile, ring=ring)), F.conv2d(x, w, b, padding=1))
    elif kind in ("shortcut_two_src", "shortcut_splitk"):
        h = bf(torch.randn(8, 128, 32, 2, generator=g))
        xa, xb = bf(torch.randn(8, 128, 32, 2, generator=g)), bf(torch.randn(8, 64, 32, 2, generator=g))
        w, b = bf(torch.randn(128, 128, 3, 3, generator=g) * 0.04), torch.randn(128, generator=g)
        ws_, bs_ = bf(torch.randn(128, 192, 1, 1, generator=g) * 0.06), torch.randn(128, generator=g)
        want = F.conv2d(h, w, b, padding=1) + F.conv2d(torch.cat([xa, xb], 1), ws_, bs_)
        pw = ops.pack_conv_shortcut(w.to(DEV), b.to(DEV), ws_.to(DEV), bs_.to(DEV))
        for sp in ((1,) if kind == "shortcut_two_src" else (3, 7, 21)):    # 18 + 3 K-tiles: boundaries inside the taps, at and inside the segment
            y = ops.conv(nhwc(h), pw, pad=(1, 1), x3=nhwc(xa), x4=nhwc(xb), tile=tile, ring=ring, splits=sp)
            close(to_nchw(y), want)
    else:
        x, x2 = bf(torch.randn(8, 256, 63, 4, generator=g)), bf(torch.randn(8, 128, 63, 4, generator=g))
        w, b = bf(torch.randn(256, 384, 1, 1, generator=g) * 0.05), torch.randn(256, generator=g)
        y = ops.conv(nhwc(x), ops.pack_conv(w.to(DEV), b.to(DEV)), x2=nhwc(x2), tile=tile, ring=ring)
        close(to_nchw(y), F.conv2d(torch.cat([x, x2], 1), w, b))


@pytest.mark.parametrize("kind", ["plain", "two_src_res", "upsample", "splitk", "act_out2", "conv1d_dil"])
def test_conv_wave_specialised_tile(ops, kind):
    """tile 12 (csrc/igemm_ws.hip): 8 compute waves + 4 loader waves feeding the LDS-DMA ring -- same results as torch for every launch
    form the big-M convolutions use (two sources + residual, nearest up-sampling, split-K, activation epilogue with a second output,
    dilated 1-D taps), ragged M (not a multiple of 256) and N = 128 / 256."""
    g = torch.Generator().manual_seed(120)
    if kind == "conv1d_dil":
        x = bf(torch.randn(2, 128, 1, 1500, generator=g))
        w, b = bf(torch.randn(128, 128, 1, 7, generator=g) * 0.04), torch.randn(128, generator=g)
        want = F.conv2d(x, w, b, padding=(0, 9), dilation=(1, 3))
        y = ops.conv(nhwc(x), ops.pack_conv(w.to(DEV), b.to(DEV)), pad=(0, 9), dil=(1, 3), tile=12)
        close(to_nchw(y), want)
        return
    x = bf(torch.randn(2, 128, 37, 16, generator=g))
    w, b = bf(torch.randn(256, 128, 3, 3, generator=g) * 0.04), torch.randn(256, generator=g)
    pw = ops.pack_conv(w.to(DEV), b.to(DEV))
    if kind == "plain":
        close(to_nchw(ops.conv(nhwc(x), pw, pad=(1, 1), tile=12)), F.conv2d(x, w, b, padding=1))
    elif kind == "two_src_res":
        x1, x2 = x[:, :64].contiguous(), x[:, 64:].contiguous()
        r = bf(torch.randn(2, 256, 37, 16, generator=g))
        close(to_nchw(ops.conv(nhwc(x1), pw, x2=nhwc(x2), pad=(1, 1), res=nhwc(r), tile=12)), F.conv2d(x, w, b, padding=1) + r)
    elif kind == "upsample":
        want = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
        close(to_nchw(ops.conv(nhwc(x), pw, pad=(1, 1), up_size=(74, 32), tile=12)), want)
    elif kind == "splitk":
        close(to_nchw(ops.conv(nhwc(x), pw, pad=(1, 1), tile=12, splits=3)), F.conv2d(x, w, b, padding=1))
    else:
        out2 = torch.empty(2, 37, 16, 256, dtype=torch.bfloat16, device=DEV)
        y = ops.conv(nhwc(x), pw, pad=(1, 1), tile=12, out2=out2, post_act=ops.ACT_LRELU, post_slope=0.1)
        want = F.conv2d(x, w, b, padding=1)
        close(to_nchw(y), want)
        close(to_nchw(out2), F.leaky_relu(want, 0.1))


def test_elementwise(ops):
    g = torch.Generator().manual_seed(11)
    t = torch.tensor([996.0, 1.0, 501.0])
    e = ops.timestep_embedding(t.to(DEV), 3, 128)
    kf = torch.exp(-math.log(10000.0) * torch.arange(64).double() / 64)
    arg = t.double()[:, None] * kf[None]
    close(e, torch.cat([arg.cos(), arg.sin()], -1).float(), rtol=1e-2, atol=1e-2)
    x = torch.randn(2, 8, 5, 4, generator=g)
    y = ops.nchw_to_nhwc(x.to(DEV))
    assert torch.equal(y.float().cpu(), bf(x).permute(0, 2, 3, 1))
    z = ops.nhwc_to_nchw_f32(y)
    assert torch.equal(z.cpu(), bf(x))


@pytest.mark.parametrize("M,Cc,N,r,geglu", [(1000, 256, 768, 4, False), (300, 128, 1024, 0, True), (70, 640, 1920, 16, False)])
def test_layernorm_folded_into_gemm(ops, M, Cc, N, r, geglu):
    """LN(x) W^T (+ LoRA on LN(x)) computed from the RAW x with in-kernel row statistics."""
    g = torch.Generator().manual_seed(21)
    x = bf(torch.randn(M, Cc, generator=g) * 1.7 + 0.4)
    w = bf(torch.randn(N, Cc, generator=g) / math.sqrt(Cc))
    b = torch.randn(N, generator=g)
    gm, bt = torch.randn(Cc, generator=g) * 0.3 + 1, torch.randn(Cc, generator=g) * 0.2
    xn = F.layer_norm(x, (Cc,), gm, bt, 1e-5)
    want = xn @ w.t() + b
    pw = ops.pack_linear_ln(w.to(DEV), b.to(DEV), gm.to(DEV), bt.to(DEV), geglu=geglu)
    if r:
        A = bf(torch.randn(r, Cc, generator=g) / r)
        Bm = bf(torch.randn(N, r, generator=g) * 0.05)
        want = want + 2.0 * (xn @ A.t()) @ Bm.t()
        ops.attach_lora(pw, [(0, N, A.to(DEV), Bm.to(DEV), 2.0)])
    if geglu:
        want = want[:, :N // 2] * F.gelu(want[:, N // 2:])
    got = ops.linear(x.to(torch.bfloat16).to(DEV), pw)
    close(got, want, rtol=2e-2)


@pytest.mark.parametrize("ring", [2, 3])
def test_conv3x3_eight_wave_tile(ops, ring):
    g = torch.Generator().manual_seed(31)
    x = bf(torch.randn(2, 128, 50, 16, generator=g))
    x2 = bf(torch.randn(2, 64, 50, 16, generator=g))
    w = bf(torch.randn(192, 192, 3, 3, generator=g) / math.sqrt(9 * 192))
    b = torch.randn(192, generator=g)
    r = bf(torch.randn(2, 192, 50, 16, generator=g))
    want = F.conv2d(torch.cat([x, x2], 1), w, b, padding=1) + r
    y = ops.conv(nhwc(x), ops.pack_conv(w.to(DEV), b.to(DEV)), x2=nhwc(x2), pad=(1, 1), res=nhwc(r), tile=6, ring=ring)
    close(to_nchw(y), want)


@pytest.mark.parametrize("d,H,N,B", [(32, 8, 1000, 2), (48, 8, 252, 2), (80, 8, 64, 3), (16, 4, 40, 2)])
def test_attention_fp8_operands(ops, d, H, N, B):
    """BASELINE config 5: e4m3 Q/K/V/P MFMA operands, fp32 accumulation.  Compared (a) with the fp32 reference at an fp8-level
    tolerance and (b) with a reference whose Q/K/V are rounded to e4m3 first (isolates the kernel from the input rounding)."""
    g = torch.Generator().manual_seed(d + N)
    C = H * d
    q, k, v = (bf(torch.randn(B, N, C, generator=g)) for _ in range(3))
    qk = torch.cat([q, k], 2).reshape(B * N, 2 * C).to(torch.bfloat16).to(DEV)
    npad = (N + 7) // 8 * 8
    vt = torch.zeros(B, C, npad, dtype=torch.bfloat16, device=DEV)
    vt[:, :, :N] = v.transpose(1, 2).to(torch.bfloat16).to(DEV)
    got = ops.attention(qk, vt, B, N, H, d, fp8=True).float().cpu().view(B, N, C)
    sp = lambda t: t.view(B, N, H, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, N, C)
    q8, k8, v8 = (t.to(torch.float8_e4m3fn).float() for t in (q, k, v))
    ref8 = F.scaled_dot_product_attention(sp(q8), sp(k8), sp(v8)).transpose(1, 2).reshape(B, N, C)
    import conftest
    rel = lambda a, b_: conftest.record(float((a - b_).norm() / b_.norm()))
    assert torch.isfinite(got).all()
    assert rel(got, ref8) < 4e-2, rel(got, ref8)          # P in e4m3 (3 mantissa bits), everything else exact
    assert rel(got, ref) < 8e-2, rel(got, ref)
    bf16_out = ops.attention(qk, vt, B, N, H, d).float().cpu().view(B, N, C)
    assert rel(bf16_out, ref) < rel(got, ref)             # sanity: the bf16 kernel is the more accurate one
