"""GPU: one-shot race screen for every kernel family whose operands reach LDS asynchronously (LDS-DMA rings with padding zeros written
through the buffer descriptor's range check, register-prefetched tiles behind counted waits).  A read that beats its data shows up only
when the LDS still holds something else, so each case launches TWO different problems alternately from an idle GPU (every launch finds the
other problem's bytes in LDS, caches and clocks are cold) and compares every result bit for bit with that problem's first result.
Same shape as tests/test_gpu_pgemm.py::test_cold_launches_are_deterministic, which covers the LN + V^T + LoRA projection kind; this file
covers the convolution ring (padding by descriptor range, split-K), the halo kernel (incl. GroupNorm of the input inside the launch), the
projection kernel's residual / row-statistics / LoRA and GEGLU kinds, the flash attention kernels (d = 32 / 48 / 80 sites, the 64-token
block kernel, the wide VAE head).  Each case runs ONCE per test session.
[REF script/train/train_audioldm_lora.py:539-546] (UNet2DConditionModel.forward), [REF script/inference/generate_audio.py:47-52]."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
REPS = 40


def dv(t):
    return t.to(torch.bfloat16).to(DEV)


@pytest.fixture(scope="module")
def ops():
    from audioldm_with_lora_amd import ops as o
    return o


def alternate(run, nprob=2, reps=REPS):
    """run(k) -> tuple of tensors for problem k; every launch starts on an idle GPU"""
    first = [None] * nprob
    for rep in range(reps):
        k = rep % nprob
        torch.cuda.synchronize()
        got = tuple(t.clone() for t in run(k))
        torch.cuda.synchronize()
        if first[k] is None:
            first[k] = got
        else:
            for a, b in zip(got, first[k]):
                assert torch.equal(a, b), (k, rep, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize("tile,splits,stride", [(2, 1, 1), (4, 6, 1), (3, 1, 2), (1, 1, 1), (6, 1, 1), (12, 1, 1), (13, 3, 1), (14, 4, 1), (13, 1, 2)])   # 12 - 14: the loader-wave tiles
def test_conv_ring_padding_by_descriptor_range(ops, tile, splits, stride):
    """igemm_pipe_kernel: 3x3 taps whose padded positions are DMA'd as zeros (voffset 0x80000000), ragged M (rows past M clamped)"""
    g = torch.Generator().manual_seed(100 + tile)
    B, H, W, Cin, Cout = 3, 21, 16, 128, 192
    probs = []
    for k in range(2):
        x = dv(torch.randn(B, H, W, Cin, generator=g) * (1.0 + k))
        pw = ops.pack_conv((torch.randn(Cout, Cin, 3, 3, generator=g) / 30).to(DEV), torch.randn(Cout, generator=g).to(DEV))
        probs.append((x, pw))
    alternate(lambda k: (ops.conv(probs[k][0], probs[k][1], stride=(stride, stride), pad=(1, 1), tile=tile, splits=splits),))


@pytest.mark.parametrize("tile,up,splits", [(7, False, 1), (8, False, 1), (7, True, 1), (7, False, 2), (15, False, 1), (16, False, 2), (15, True, 1), (15, False, 2)])   # 15 / 16: loader waves
def test_halo_kernel_zero_rows(ops, tile, up, splits):
    """igemm_halo_kernel: the halo's border rows / columns are OOB lanes of the LDS-DMA (zeros through the descriptor)"""
    g = torch.Generator().manual_seed(200 + tile)
    B, H, W, C1, C2, Cout = 2, (6 if up else 24), (8 if up else 16), 64, 64, 128
    probs = []
    for k in range(2):
        x1, x2 = dv(torch.randn(B, H, W, C1, generator=g)), dv(torch.randn(B, H, W, C2, generator=g) * 2)
        pw = ops.pack_conv((torch.randn(Cout, C1 + C2, 3, 3, generator=g) / 30).to(DEV), torch.randn(Cout, generator=g).to(DEV))
        probs.append((x1, x2, pw))
    kw = dict(pad=(1, 1), tile=tile, splits=splits, up_size=((2 * H, 2 * W) if up else None))
    alternate(lambda k: (ops.conv(probs[k][0], probs[k][2], x2=probs[k][1], **kw),))


@pytest.mark.parametrize("tile,splits", [(15, 1), (16, 1), (15, 2), (16, 3)])
def test_halo_ws_fused_shortcut_segment(ops, tile, splits):
    """igemm_halo_ws_kernel, EXT: single-tap items with their own halo image in a three-buffer rotation, loaders two items ahead"""
    g = torch.Generator().manual_seed(250 + tile + splits)
    B, H, W, C, Ce, Cout = 2, 24, 16, 128, 192, 128
    probs = []
    for k in range(2):
        h, xa, xb = (dv(torch.randn(B, H, W, c, generator=g) * (1 + k)) for c in (C, 128, 64))
        pw = ops.pack_conv_shortcut((torch.randn(Cout, C, 3, 3, generator=g) / 30).to(DEV), torch.randn(Cout, generator=g).to(DEV),
                                    (torch.randn(Cout, Ce, 1, 1, generator=g) / 12).to(DEV), torch.randn(Cout, generator=g).to(DEV))
        probs.append((h, xa, xb, pw))
    alternate(lambda k: (ops.conv(probs[k][0], probs[k][3], pad=(1, 1), x3=probs[k][1], x4=probs[k][2], tile=tile, ring=3, splits=splits),))


@pytest.mark.parametrize("tile", [0, 15, 16])
def test_halo_kernel_groupnorm_of_the_input(ops, tile):
    """GNIN instantiation: the chunk is normalised in place between its vmcnt wait and the barrier that releases it (tiles 15 / 16: by
    the loader waves)"""
    g = torch.Generator().manual_seed(7)
    B, H, W, C, N = 2, 40, 16, 128, 128
    probs = []
    for k in range(2):
        x0 = dv(torch.randn(B, H, W, C, generator=g) * (1 + k))
        y = ops.conv(x0, ops.pack_conv((torch.randn(C, C, 3, 3, generator=g) / 30).to(DEV), None), pad=(1, 1), splits=1, qstats=True)
        ops.QSTATS_MIN_HW, keep = 1, ops.QSTATS_MIN_HW
        try:
            y = ops.conv(x0, ops.pack_conv((torch.randn(C, C, 3, 3, generator=g) / 30).to(DEV), None), pad=(1, 1), splits=1, qstats=True)
        finally:
            ops.QSTATS_MIN_HW = keep
        pw = ops.pack_conv((torch.randn(N, C, 3, 3, generator=g) / 30).to(DEV), torch.randn(N, generator=g).to(DEV))
        gm, bt = (torch.randn(C, generator=g) * 0.3 + 1).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV)
        probs.append((y, pw, gm, bt))
    if not ops.gn_in_ok(probs[0][0], None, probs[0][1], (1, 1), (1, 1), (1, 1), None, None):
        pytest.skip("this geometry does not take the input-norm fold")
    alternate(lambda k: (ops.conv(probs[k][0], probs[k][1], pad=(1, 1), tile=tile, gn_in=(probs[k][2], probs[k][3], 32, 1e-5, 1)),))


@pytest.mark.parametrize("M,K,N,r", [(2016, 384, 384, 4), (512, 640, 640, 4), (8000, 256, 256, 4)])
def test_pgemm_residual_rowstats_lora(ops, M, K, N, r):
    """pgemm RES | RSTAT + LoRA kind (the out-projection of an Attention module): residual tiles and LoRA-B rows ride the ring"""
    g = torch.Generator().manual_seed(300 + K)
    probs = []
    for k in range(2):
        x, res = dv(torch.randn(M, K, generator=g)), dv(torch.randn(M, N, generator=g))
        pw = ops.pack_linear((torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV), torch.randn(N, generator=g).to(DEV))
        ops.attach_lora(pw, [(0, N, (torch.randn(r, K, generator=g) / r).to(DEV), (torch.randn(N, r, generator=g) * 0.05).to(DEV), 2.0)])
        probs.append((x, res, pw))
    alternate(lambda k: ops.linear(probs[k][0], probs[k][2], res=probs[k][1], rowstats=True))


@pytest.mark.parametrize("M,K", [(8000, 256), (2016, 384)])
def test_pgemm_geglu_with_folded_layernorm(ops, M, K):
    g = torch.Generator().manual_seed(400 + K)
    N = 8 * K
    probs = []
    for k in range(2):
        x = dv(torch.randn(M, K, generator=g) * 1.5 + 0.3)
        pw = ops.pack_linear_ln((torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV), torch.randn(N, generator=g).to(DEV),
                                (torch.randn(K, generator=g) * 0.3 + 1).to(DEV), (torch.randn(K, generator=g) * 0.2).to(DEV), geglu=True)
        xs = x.float().view(M, 4, K // 4)
        parts = torch.stack([xs.sum(2), (xs * xs).sum(2)], dim=2).contiguous()
        probs.append((x, pw, parts))
    alternate(lambda k: (ops.linear(probs[k][0], probs[k][1], ln_parts=probs[k][2]),))


@pytest.mark.parametrize("B,N,H,d", [(8, 1000, 8, 32), (8, 252, 8, 48), (8, 64, 8, 80)])
def test_attention_sites(ops, B, N, H, d):
    g = torch.Generator().manual_seed(500 + d)
    C = H * d
    npad = (N + 7) // 8 * 8
    probs = []
    for k in range(2):
        qk = dv(torch.randn(B * N, 2 * C, generator=g) * (0.5 + 0.5 * k))
        vt = torch.zeros(B, C, npad, dtype=torch.bfloat16, device=DEV)
        vt[:, :, :N] = dv(torch.randn(B, C, N, generator=g))
        probs.append((qk, vt))
    alternate(lambda k: (ops.attention(probs[k][0], probs[k][1], B, N, H, d, prescaled=True),))


def test_attn_block64(ops):
    g = torch.Generator().manual_seed(64)
    B, N, H, d, r = 8, 64, 8, 80, 4
    Cc = H * d
    probs = []
    for k in range(2):
        x = dv(torch.randn(B * N, Cc, generator=g) * 1.3 + 0.2)
        pw = ops.pack_linear_ln((torch.randn(3 * Cc, Cc, generator=g) / math.sqrt(Cc)).to(DEV), None,
                                (torch.randn(Cc, generator=g) * 0.3 + 1).to(DEV), (torch.randn(Cc, generator=g) * 0.2).to(DEV))
        ops.attach_lora(pw, [(i * Cc, Cc, (torch.randn(r, Cc, generator=g) / math.sqrt(Cc)).to(DEV),
                              (torch.randn(Cc, r, generator=g) * 0.3).to(DEV), 2.0) for i in range(3)])
        xs = x.float().view(B * N, Cc // 64, 64)
        parts = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).contiguous()
        probs.append((x, pw, parts))
    assert ops.attn_block64_ok(probs[0][1], N, H, d, probs[0][2])
    alternate(lambda k: (ops.attn_block64(probs[k][0], probs[k][1], probs[k][2], B, N, H, d),))


def test_attn_block256(ops, monkeypatch):
    monkeypatch.setattr(ops, "ATTN_BLOCK256", True)
    g = torch.Generator().manual_seed(256)
    B, N, H, d, r = 8, 252, 8, 48, 4
    Cc = H * d
    probs = []
    for k in range(2):
        x = dv(torch.randn(B * N, Cc, generator=g) * 1.3 + 0.2)
        pw = ops.pack_linear_ln((torch.randn(3 * Cc, Cc, generator=g) / math.sqrt(Cc)).to(DEV), None,
                                (torch.randn(Cc, generator=g) * 0.3 + 1).to(DEV), (torch.randn(Cc, generator=g) * 0.2).to(DEV))
        ops.attach_lora(pw, [(i * Cc, Cc, (torch.randn(r, Cc, generator=g) / math.sqrt(Cc)).to(DEV),
                              (torch.randn(Cc, r, generator=g) * 0.3).to(DEV), 2.0) for i in range(3)])
        xs = x.float().view(B * N, Cc // 64, 64)
        parts = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).contiguous()
        probs.append((x, pw, parts))
    assert ops.attn_block_ok(probs[0][1], N, H, d, probs[0][2]) == 256
    alternate(lambda k: (ops.attn_block(probs[k][0], probs[k][1], probs[k][2], B, N, H, d),))


def test_attention_wide(ops):
    """attention_wide_kernel: K / V^T tiles by LDS-DMA, keys zero-padded to a multiple of 32 through the descriptor"""
    g = torch.Generator().manual_seed(512)
    B, N, d = 2, 2000, 512
    npad = (N + 31) // 32 * 32
    probs = []
    for k in range(2):
        qk = dv(torch.randn(B * N, 2 * d, generator=g) * (0.2 + 0.1 * k))
        vt = torch.zeros(B, d, npad, dtype=torch.bfloat16, device=DEV)
        vt[:, :, :N] = dv(torch.randn(B, d, N, generator=g))
        probs.append((qk, vt))
    alternate(lambda k: (ops.attention_wide(probs[k][0], probs[k][1], B, N, d),), reps=24)
