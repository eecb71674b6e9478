"""GPU parity of the backward kernels vs torch autograd on the CPU (fp32) with bf16-rounded inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16).float()


def close(got, want, rtol=2e-2, atol=None):
    got, want = got.float().cpu(), want.float()
    assert got.shape == want.shape, (got.shape, want.shape)
    atol = atol if atol is not None else 1.5e-2 * float(want.abs().max()) + 1e-7
    err = (got - want).abs()
    bad = ~(err <= atol + rtol * want.abs())
    assert not bad.any(), f"max err {float(err.max()):.4g} ref max {float(want.abs().max()):.4g} bad {int(bad.sum())}"


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2)


@pytest.fixture(scope="module")
def ops():
    from audioldm_with_lora_amd import ops as o
    return o


@pytest.mark.parametrize("B,HW,C1,C2,groups,act", [(2, (16, 16), 128, 0, 32, 1), (2, (7, 4), 96, 64, 8, 1), (1, (63, 4), 384, 256, 32, 0)])
def test_groupnorm_bwd(ops, B, HW, C1, C2, groups, act):
    g = torch.Generator().manual_seed(0)
    H, W = HW
    x = bf(torch.randn(B, C1 + C2, H, W, generator=g) * 2 + 0.3).requires_grad_()
    gm, bt = torch.randn(C1 + C2, generator=g), torch.randn(C1 + C2, generator=g)
    dy = bf(torch.randn(B, C1 + C2, H, W, generator=g))
    y = F.group_norm(x, groups, gm, bt, 1e-5)
    if act:
        y = F.silu(y)
    y.backward(dy)
    x1, x2 = nhwc(x.detach()[:, :C1]), (nhwc(x.detach()[:, C1:]) if C2 else None)
    dx, dx2 = ops.groupnorm_bwd(x1, nhwc(dy), gm.to(DEV), bt.to(DEV), groups, 1e-5, act, x2=x2)
    close(nchw(dx), x.grad[:, :C1])
    if C2:
        close(nchw(dx2), x.grad[:, C1:])
    # accumulation of gradients the tensors already hold (residual / skip joins), out of place
    p1 = bf(torch.randn(B, C1, H, W, generator=g))
    p2 = bf(torch.randn(B, C2, H, W, generator=g)) if C2 else None
    b1, b2 = nhwc(p1), (nhwc(p2) if C2 else None)
    keep1 = b1.clone()
    r1, r2 = ops.groupnorm_bwd(x1, nhwc(dy), gm.to(DEV), bt.to(DEV), groups, 1e-5, act, x2=x2, dx_add=b1, dx2_add=b2)
    assert r1.data_ptr() != b1.data_ptr() and torch.equal(b1, keep1)          # the prior gradient is not modified
    close(nchw(r1), x.grad[:, :C1] + p1)
    if C2:
        close(nchw(r2), x.grad[:, C1:] + p2)


def test_layernorm_and_geglu_bwd(ops):
    g = torch.Generator().manual_seed(1)
    for Cc in (64, 256, 640):
        x = bf(torch.randn(50, Cc, generator=g) * 2 + 1).requires_grad_()
        gm, bt = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
        dy = bf(torch.randn(50, Cc, generator=g))
        F.layer_norm(x, (Cc,), gm, bt, 1e-5).backward(dy)
        dx = ops.layernorm_bwd(x.detach().to(torch.bfloat16).to(DEV), dy.to(torch.bfloat16).to(DEV), gm.to(DEV))
        close(dx, x.grad)
        prev = bf(torch.randn(50, Cc, generator=g))
        buf = prev.to(torch.bfloat16).to(DEV)
        out = ops.layernorm_bwd(x.detach().to(torch.bfloat16).to(DEV), dy.to(torch.bfloat16).to(DEV), gm.to(DEV), dx_add=buf)
        assert out.data_ptr() != buf.data_ptr()
        close(out, x.grad + prev)
    # GEGLU on the interleaved layout
    M, I = 40, 64
    h = bf(torch.randn(M, 2 * I, generator=g)).requires_grad_()
    dout = bf(torch.randn(M, I, generator=g))
    idx = torch.arange(I).view(-1, 16)
    order = torch.cat([idx, idx + I], 1).reshape(-1)            # packed position -> original row
    inv = torch.empty_like(order); inv[order] = torch.arange(2 * I)
    y = h[:, :I] * F.gelu(h[:, I:])
    y.backward(dout)
    hp = h.detach()[:, order].contiguous().to(torch.bfloat16).to(DEV)
    close(ops.geglu_fwd(hp), y.detach())
    dh = ops.geglu_bwd(hp, dout.to(torch.bfloat16).to(DEV))
    close(dh.float().cpu()[:, inv], h.grad)


def test_conv_dx_via_igemm(ops):
    g = torch.Generator().manual_seed(2)
    # stride-1 3x3 with two sources
    x = bf(torch.randn(2, 160, 9, 8, generator=g)).requires_grad_()
    w = bf(torch.randn(96, 160, 3, 3, generator=g) / 30)
    dy = bf(torch.randn(2, 96, 9, 8, generator=g))
    F.conv2d(x, w, None, padding=1).backward(dy)
    d1 = ops.conv(nhwc(dy), ops.pack_conv_bwd(w.to(DEV), 0, 96), pad=(1, 1))
    d2 = ops.conv(nhwc(dy), ops.pack_conv_bwd(w.to(DEV), 96, 160), pad=(1, 1))
    close(nchw(d1), x.grad[:, :96]); close(nchw(d2), x.grad[:, 96:])
    # stride-2 3x3 (down-sampler): zero-dilated gather
    for (H, W) in ((16, 8), (63, 4)):
        x = bf(torch.randn(2, 64, H, W, generator=g)).requires_grad_()
        w = bf(torch.randn(64, 64, 3, 3, generator=g) / 24)
        y = F.conv2d(x, w, None, stride=2, padding=1)
        dy = bf(torch.randn(y.shape, generator=g))
        y.backward(dy)
        dx = ops.conv(nhwc(dy), ops.pack_conv_bwd(w.to(DEV)), pad=(1, 1), in_dilate=2, out_hw=(H, W))
        close(nchw(dx), x.grad)
    # nearest up-sample + conv
    for (ih, iw, oh, ow) in ((8, 4, 16, 8), (32, 2, 63, 4)):
        x = bf(torch.randn(1, 64, ih, iw, generator=g)).requires_grad_()
        w = bf(torch.randn(64, 64, 3, 3, generator=g) / 24)
        y = F.conv2d(F.interpolate(x, size=(oh, ow), mode="nearest"), w, None, padding=1)
        dy = bf(torch.randn(y.shape, generator=g))
        y.backward(dy)
        dup = ops.conv(nhwc(dy), ops.pack_conv_bwd(w.to(DEV)), pad=(1, 1))
        close(nchw(ops.upsample_nearest_bwd(dup, ih, iw)), x.grad)


# the last three rows are config 3's real attention sites (8 heads: N = 1024 / d = 32, N = 256 / d = 48 on the split-key path,
# N = 64 / d = 80), batch 2
@pytest.mark.parametrize("B,N,H,d", [(2, 200, 4, 32), (1, 252, 8, 48), (2, 64, 4, 80), (1, 1000, 2, 32), (1, 40, 4, 24),
                                     (2, 1024, 8, 32), (2, 256, 8, 48), (2, 64, 8, 80)])
def test_attention_fwd_lse_and_bwd(ops, B, N, H, d):
    g = torch.Generator().manual_seed(3)
    Cc = H * d
    qkv = bf(torch.randn(B * N, 3 * Cc, generator=g)).requires_grad_()
    dO = bf(torch.randn(B * N, Cc, generator=g))
    sp = lambda z: z.reshape(B, N, H, d).transpose(1, 2)
    q, k, v = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B * N, Cc)
    o.backward(dO)
    dev = qkv.detach().to(torch.bfloat16).to(DEV)
    qkvT = ops.transpose_tokens(dev, B, N, 3 * Cc)
    assert torch.equal(qkvT[:, :, :N].cpu(), dev.cpu().view(B, N, 3 * Cc).transpose(1, 2))
    out, lse = ops.attention_train(dev, qkvT, B, N, H, d)
    close(out, o.detach(), rtol=2e-2, atol=1e-2)
    s = (sp(q) @ sp(k).transpose(-1, -2)).detach() / math.sqrt(d)
    close(lse, torch.logsumexp(s, -1) * 1.4426950408889634, rtol=1e-2)
    dqkv = ops.attention_bwd(dev, qkvT, dO.to(torch.bfloat16).to(DEV), out, lse, B, N, H, d)
    close(dqkv, qkv.grad, rtol=3e-2)


def test_tn_small_lora_pack_mse(ops):
    import ctypes
    g = torch.Generator().manual_seed(4)
    M, Rp, Qc = 300, 32, 160
    P = bf(torch.randn(M, Rp, generator=g))
    Q = bf(torch.randn(M, Qc, generator=g))
    want = P.t() @ Q                                     # [Rp][Qc]
    flat = torch.zeros(8 * Qc + 4 * Qc, device=DEV)
    # rows 0..7 -> dense rows of a [8][Qc] block ; rows 8..11 -> transposed into a [Qc][4] block, scaled by 2
    rows = torch.zeros(Rp, 24, dtype=torch.uint8)        # sizeof(TnRow) = 24: {float* dst; int qlo, qhi, qstride; float scale}
    import struct
    base = flat.data_ptr()
    for p in range(12):
        if p < 8:
            rec = struct.pack("<qiiif", base + 4 * p * Qc, 0, Qc, 1, 1.0)
        else:
            rec = struct.pack("<qiiif", base + 4 * (8 * Qc + (p - 8)), 0, Qc, 4, 2.0)
        assert len(rec) == 24
        rows[p] = torch.tensor(list(rec), dtype=torch.uint8)
    ops.tn_small(P.to(torch.bfloat16).to(DEV), Q.to(torch.bfloat16).to(DEV), rows.to(DEV))
    got = flat.cpu()
    close(got[:8 * Qc].view(8, Qc), want[:8], rtol=1e-3, atol=1e-3 * float(want.abs().max()))
    close(got[8 * Qc:].view(Qc, 4), 2 * want[8:12].t(), rtol=1e-3, atol=2e-3 * float(want.abs().max()))
    # mse
    pred, tgt = torch.randn(2, 5, 4, 8, generator=g), torch.randn(2, 5, 4, 8, generator=g)
    loss = torch.zeros(1, device=DEV)
    dp = ops.mse_grad(pred.to(DEV), tgt.to(DEV), loss)
    assert abs(float(loss) - float(F.mse_loss(pred, tgt))) < 1e-5
    close(dp, 2 * (pred - tgt) / pred.numel(), rtol=1e-2, atol=1e-5)


@pytest.mark.parametrize("M,Rp,Qc,ldq", [(8192, 32, 768, 768), (2048, 64, 384, 1152), (515, 32, 136, 136), (64, 64, 1920, 1920),
                                         (300, 32, 100, 100), (97, 32, 36, 44)])
def test_tn_small_mfma_and_scalar_paths(ops, M, Rp, Qc, ldq):
    """out[p][q] = sum_m P[m][p] Q[m][q] through the row table: aligned shapes take the MFMA / transposing-LDS-read kernel,
    ldq or Qc not a multiple of 8 the scalar one; windows (qlo, qhi), transposed scatter and scaling as the trainer uses."""
    import struct
    g = torch.Generator().manual_seed(M + Qc)
    P = bf(torch.randn(M, Rp, generator=g))
    Qfull = bf(torch.randn(M, ldq, generator=g))
    want = P.t() @ Qfull[:, :Qc]                            # [Rp][Qc]
    half = (Qc // 2) // 4 * 4
    flat = torch.zeros(Rp * Qc + 16, device=DEV)
    rows = torch.zeros(Rp, 24, dtype=torch.uint8)
    base = flat.data_ptr()
    nrows = Rp - 3                                          # the last rank rows are padding (dst = NULL)
    for p_ in range(nrows):
        if p_ % 2 == 0:                                     # dense row, full window
            rec = struct.pack("<qiiif", base + 4 * p_ * Qc, 0, Qc, 1, 1.0)
        else:                                               # upper half window only, scaled
            rec = struct.pack("<qiiif", base + 4 * p_ * Qc, half, Qc, 1, 0.5)
        rows[p_] = torch.tensor(list(rec), dtype=torch.uint8)
    Qd = Qfull.to(torch.bfloat16).to(DEV)
    ops.tn_small(P.to(torch.bfloat16).to(DEV), Qd, rows.to(DEV), Qc=Qc)
    got = flat.cpu()[:Rp * Qc].view(Rp, Qc)
    tol = 2e-3 * float(want.abs().max())
    for p_ in range(Rp):
        if p_ >= nrows:
            assert float(got[p_].abs().max()) == 0.0
        elif p_ % 2 == 0:
            close(got[p_], want[p_], rtol=1e-3, atol=tol)
        else:
            close(got[p_, :Qc - half], 0.5 * want[p_, half:], rtol=1e-3, atol=tol)
            assert float(got[p_, Qc - half:].abs().max()) == 0.0
    assert float(flat[Rp * Qc:].abs().max()) == 0.0        # nothing written past the table's windows


@pytest.mark.parametrize("B,N,C,ld", [(2, 100, 96, 96), (8, 64, 640, 1920), (1, 1024, 256, 768), (3, 37, 40, 40), (2, 50, 20, 20), (2, 9, 12, 36)])
def test_transpose_tokens_fast_and_generic_paths(ops, B, N, C, ld):
    """[B*N][ld] row-major (first C columns) -> [B][C][Npad] token-major with zero padding; C, ld multiples of 8 take the
    transposing-LDS-read kernel, anything else the scalar one."""
    g = torch.Generator().manual_seed(N + C)
    full = torch.randn(B * N, ld, generator=g).to(torch.bfloat16)
    npad = (N + 7) // 8 * 8
    dev = full.to(DEV)
    got = ops.transpose_tokens(dev[:, :C] if ld != C else dev, B, N, C).cpu()
    assert got.shape == (B, C, npad)
    want = torch.zeros(B, C, npad, dtype=torch.bfloat16)
    want[:, :, :N] = full[:, :C].view(B, N, C).transpose(1, 2)
    assert torch.equal(got, want)


def test_tn_batched_many_jobs_one_launch(ops):
    """ops.TnBatch: several (P, Q) products of different shapes and rank paddings in one launch per Rp == the per-call path."""
    import struct
    g = torch.Generator().manual_seed(77)
    shapes = [(1024, 32, 256, 256), (300, 32, 96, 288), (64, 64, 1920, 1920), (2048, 32, 384, 384), (515, 64, 136, 136), (97, 32, 36, 44)]
    tb = ops.TnBatch(16, DEV)
    outs, wants, keep = [], [], []
    for (M, Rp, Qc, ldq) in shapes:
        P = bf(torch.randn(M, Rp, generator=g))
        Q = bf(torch.randn(M, ldq, generator=g))
        flat = torch.zeros(Rp * Qc, device=DEV)
        rows = torch.zeros(Rp, 24, dtype=torch.uint8)
        for p_ in range(Rp):
            rows[p_] = torch.tensor(list(struct.pack("<qiiif", flat.data_ptr() + 4 * p_ * Qc, 0, Qc, 1, 1.0)), dtype=torch.uint8)
        Pd, Qd, rd = P.to(torch.bfloat16).to(DEV), Q.to(torch.bfloat16).to(DEV), rows.to(DEV)
        tb.add(Pd, Qd[:, :Qc] if ldq != Qc else Qd, rd, Qc)          # (97, 32, 36, 44): ldq % 8 != 0 -> immediate scalar path
        keep.append((Pd, Qd, rd))
        outs.append(flat)
        wants.append(P.t() @ Q[:, :Qc])
    assert len(tb.jobs[32]) == 3 and len(tb.jobs[64]) == 2
    tb.launch()
    assert not tb.jobs[32] and not tb.jobs[64] and not tb.keep
    for flat, want in zip(outs, wants):
        close(flat.cpu().view(want.shape), want, rtol=1e-3, atol=2e-3 * float(want.abs().max()))


@pytest.mark.parametrize("B,C1,C2,Cout,H,W,splits,act", [(8, 640, 0, 640, 16, 4, 6, 1), (2, 256, 128, 384, 63, 4, 3, 1), (2, 128, 0, 64, 25, 16, 2, 0)])
def test_groupnorm_bwd_takes_dy_as_split_partials(B, C1, C2, Cout, H, W, splits, act):
    """dX conv with defer -> groupnorm_bwd sums the partial tiles: bit-identical to reduce launch + groupnorm_bwd"""
    from audioldm_with_lora_amd import ops
    g = torch.Generator().manual_seed(23)
    C = C1 + C2
    x = (torch.randn(B, H, W, C1, generator=g) * 1.5 + 0.2).to(torch.bfloat16).cuda()
    x2 = (torch.randn(B, H, W, C2, generator=g)).to(torch.bfloat16).cuda() if C2 else None
    gm, bt = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    dyo = torch.randn(B, H, W, Cout, generator=g).to(torch.bfloat16).cuda()              # gradient of the conv's OUTPUT
    wt = (torch.randn(C, Cout, 3, 3, generator=g) / (3 * Cout ** 0.5)).cuda()            # transposed filter of the dX conv
    pw = ops.pack_conv(wt, None)
    ref_dy = ops.conv(dyo, pw, pad=(1, 1), splits=splits)
    want = ops.groupnorm_bwd(x, ref_dy, gm, bt, 32, 1e-5, act, x2=x2)
    d = ops.conv(dyo, pw, pad=(1, 1), splits=splits, defer=(C, 32))
    assert isinstance(d, ops.Deferred)
    got = ops.groupnorm_bwd(x, d, gm, bt, 32, 1e-5, act, x2=x2)
    assert torch.equal(got[0], want[0])
    if C2:
        assert torch.equal(got[1], want[1])
    assert not ops._PENDING_BY_STREAM
