"""GPU parity of the transformer blocks' projection GEMM (csrc/pgemm.hip, aldm_pgemm) against a torch-CPU fp32 reference of the
same op on bf16-rounded inputs, through ops.conv / ops.linear (the route the UNet takes), for every epilogue the blocks use:
plain (+ bias), residual, LoRA, LayerNorm folded (statistics handed over), V^T store, GEGLU, row statistics; every K the kernel
is built for; every launch shape (rows per workgroup, tile width, tiles per range); ragged M; token counts per sample that are
multiples of 8, of 4 only, and odd (the three V^T store paths).
[REF script/train/train_audioldm_lora.py:539-546] (UNet2DConditionModel.forward), [REF train:378-385] (peft LoRA)."""
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
    bad = ~(err <= atol + rtol * want.abs())
    assert not bad.any(), f"max err {float(err.max()):.4g} (ref max {float(want.abs().max()):.4g}), {int(bad.sum())} bad"


def dv(t):
    return t.to(torch.bfloat16).to(DEV)


@pytest.fixture(scope="module")
def ops():
    from audioldm_with_lora_amd import ops as o
    assert o.PGEMM, "ALDM_NO_PGEMM is set: these tests need the pgemm route"
    return o


@pytest.fixture()
def labels(ops):
    """records the launch labels of a test so that it can assert WHICH kernel family ran"""
    ops.PROFILE = []
    yield ops.PROFILE
    ops.PROFILE = None


def shapes_for(K, N, geglu=False, vt_col0=None, res=False, rp=0):
    """every (mi, nt, tiles_per_range, waves) the kernel accepts for this GEMM (160 KiB of LDS: ring + vectors + LoRA-B rows),
    thinned to a handful"""
    out = []
    for nw in (4, 8):
        for mi in ((1, 2) if ((K <= 384 or rp == 0) and nw == 4) else (1,)):
            for nt in ((64,) if geglu else (32, 64)):
                if vt_col0 is not None and vt_col0 % nt:
                    continue
                nti = N // nt
                bm = 16 * mi * nw
                stage = nt * K * 2 + (bm * nt * 2 if res else 0)
                tprs = [t for t in range(1, nti + 1) if nti % t == 0 and t * nt <= 512
                        and min(t, 3) * stage + 2 * bm * 4 + 2 * t * nt * 4 + t * nt * rp * 2 <= 160 * 1024]
                if not tprs:
                    continue
                for tpr in sorted({tprs[0], tprs[len(tprs) // 2], tprs[-1]}):
                    out.append((mi, nt, tpr, nw))
    return out


def stats_of(y):
    y = y.float().cpu()
    return y.sum(1), (y * y).sum(1)


@pytest.mark.parametrize("M,K,N", [(1000, 256, 256), (2016, 384, 384), (70, 640, 640), (129, 256, 768)])
def test_plain_bias_residual_rowstats(ops, labels, M, K, N):
    g = torch.Generator().manual_seed(1)
    x = bf(torch.randn(M, K, generator=g))
    w = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g)
    res = bf(torch.randn(M, N, generator=g))
    pw = ops.pack_linear(w.to(DEV), b.to(DEV))
    for cfg in shapes_for(K, N, res=True):
        ops.PGEMM_CFG[(M, N, K, "")] = cfg
        ops.PGEMM_CFG[(M, N, K, "rs")] = cfg
        y = ops.linear(dv(x), pw)
        close(y, x @ w.t() + b)
        y, st = ops.linear(dv(x), pw, res=dv(res), rowstats=True)
        close(y, x @ w.t() + b + res)
        nr = N // (cfg[1] * cfg[2])                           # (more than 16 partial pairs: the table entry is set aside for the plan)
        assert st.shape == (M, nr, 2) if nr <= 16 else st.shape[1] <= 16
        s1, s2 = stats_of(y)                                  # the statistics are those of the values AS STORED
        assert torch.allclose(st[:, :, 0].sum(1).cpu(), s1, rtol=1e-4, atol=1e-2)
        assert torch.allclose(st[:, :, 1].sum(1).cpu(), s2, rtol=1e-4, atol=1e-2)
    ops.PGEMM_CFG.clear()
    assert labels and all(l[0].startswith("pgemm_") for l in labels), [l[0] for l in labels]


@pytest.mark.parametrize("M,K,N,r", [(1000, 256, 256, 4), (504, 384, 384, 8), (512, 640, 640, 4), (200, 256, 256, 16), (333, 384, 384, 32)])
def test_lora_residual(ops, labels, M, K, N, r):
    """peft lora.Linear with the adapter fused: y = x W^T + b + s (x A^T) B^T + res; B = 0 reproduces the base GEMM bit for bit."""
    g = torch.Generator().manual_seed(2)
    x = bf(torch.randn(M, K, generator=g))
    w = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g)
    A = bf(torch.randn(r, K, generator=g) / r)
    Bm = bf(torch.randn(N, r, generator=g) * 0.05)
    res = bf(torch.randn(M, N, generator=g))
    want = x @ w.t() + b + 2.0 * (x @ A.t()) @ Bm.t() + res
    pw = ops.pack_linear(w.to(DEV), b.to(DEV))
    ops.attach_lora(pw, [(0, N, A.to(DEV), Bm.to(DEV), 2.0)])
    pw0 = ops.pack_linear(w.to(DEV), b.to(DEV))
    pwz = ops.pack_linear(w.to(DEV), b.to(DEV))
    ops.attach_lora(pwz, [(0, N, A.to(DEV), torch.zeros_like(Bm).to(DEV), 2.0)])
    for cfg in shapes_for(K, N, res=True, rp=pw.Rp):
        for kind in ("r", f"rl{pw.Rp}s"):
            ops.PGEMM_CFG[(M, N, K, kind)] = cfg
        y, st = ops.linear(dv(x), pw, res=dv(res), rowstats=True)
        close(y, want)
        s1, _ = stats_of(y)
        assert torch.allclose(st[:, :, 0].sum(1).cpu(), s1, rtol=1e-4, atol=1e-2)
        assert torch.equal(ops.linear(dv(x), pw0, res=dv(res)), ops.linear(dv(x), pwz, res=dv(res)))
    ops.PGEMM_CFG.clear()
    assert all(l[0].startswith("pgemm_") for l in labels), [l[0] for l in labels]


def _ln_parts(ops, x, nparts):
    """row statistics as a producer GEMM would hand them over: nparts partial (sum, sum of squares) per row"""
    M, K = x.shape
    xs = x.view(M, nparts, K // nparts)
    return torch.stack([xs.sum(2), (xs * xs).sum(2)], dim=2).contiguous().to(DEV)


@pytest.mark.parametrize("B,N,C,r,nparts", [(2, 1000, 256, 4, 4), (3, 252, 384, 4, 6), (2, 64, 640, 4, 10), (2, 63, 256, 8, 2), (1, 250, 384, 0, 3)])
def test_qkv_layernorm_folded_lora_vt(ops, labels, B, N, C, r, nparts):
    """the fused to_q | to_k | to_v GEMM of an Attention module: LayerNorm folded (statistics from ln_parts), LoRA on q, k and v,
    Q | K row-major, V token-major."""
    g = torch.Generator().manual_seed(3)
    M = B * N
    x = bf(torch.randn(M, C, generator=g) * 1.7 + 0.4)
    w = bf(torch.randn(3 * C, C, generator=g) / math.sqrt(C))
    b = torch.randn(3 * C, generator=g)
    gm, bt = torch.randn(C, generator=g) * 0.3 + 1, torch.randn(C, generator=g) * 0.2
    xn = F.layer_norm(x, (C,), gm, bt, 1e-5)
    want = xn @ w.t() + b
    pw = ops.pack_linear_ln(w.to(DEV), b.to(DEV), gm.to(DEV), bt.to(DEV))
    parts = []
    for i in range(3 if r else 0):
        A = bf(torch.randn(r, C, generator=g) / r)
        Bm = bf(torch.randn(C, r, generator=g) * 0.05)
        want[:, i * C:(i + 1) * C] += 1.5 * (xn @ A.t()) @ Bm.t()
        parts.append((i * C, C, A.to(DEV), Bm.to(DEV), 1.5))
    ops.attach_lora(pw, parts)
    if pw.Rp and pw.ranks_used > 32:
        pytest.skip("combined rank > 32 stays on aldm_igemm")
    lp = _ln_parts(ops, x, nparts)
    npad = (N + 7) // 8 * 8
    for cfg in shapes_for(C, 3 * C, vt_col0=2 * C, rp=pw.Rp):
        ops.PGEMM_CFG[(M, 3 * C, C, "v" + (f"l{pw.Rp}" if pw.Rp else ""))] = cfg
        vt = torch.zeros(B, C, npad, dtype=torch.bfloat16, device=DEV)
        qk = ops.conv(dv(x).view(B, 1, N, C), pw, vt=vt, vt_col0=2 * C, vt_ld=npad, vt_batch_stride=C * npad, ln_parts=lp).view(M, 2 * C)
        close(qk, want[:, :2 * C], rtol=2e-2)
        close(vt[:, :, :N].permute(0, 2, 1).reshape(M, C), want[:, 2 * C:], rtol=2e-2)
        assert not vt[:, :, N:].any(), "the padding columns of V^T must stay untouched"
    ops.PGEMM_CFG.clear()
    assert all(l[0].startswith("pgemm_") for l in labels), [l[0] for l in labels]


@pytest.mark.parametrize("B,N,C,Nout,r", [(2, 1024, 256, 768, 8), (2, 256, 384, 1152, 4), (3, 64, 640, 640, 8), (2, 60, 256, 256, 16)])
def test_dual_store_row_major_and_token_major(ops, labels, B, N, C, Nout, r):
    """the trainer's q | k | v forward and out-projection dX launches (training.t_lora_linear): every column stored row-major AND
    token-major, LoRA fused, T = x A^T handed out (vt_dual, as aldm_igemm's): both layouts hold the same bf16 values."""
    g = torch.Generator().manual_seed(9)
    M = B * N
    x = bf(torch.randn(M, C, generator=g))
    w = bf(torch.randn(Nout, C, generator=g) / math.sqrt(C))
    b = torch.randn(Nout, generator=g)
    A = bf(torch.randn(r, C, generator=g) / r)
    Bm = bf(torch.randn(Nout, r, generator=g) * 0.05)
    want = x @ w.t() + b + 2.0 * (x @ A.t()) @ Bm.t()
    pw = ops.pack_linear(w.to(DEV), b.to(DEV))
    ops.attach_lora(pw, [(0, Nout, A.to(DEV), Bm.to(DEV), 2.0)])
    npad = (N + 7) // 8 * 8
    for cfg in shapes_for(C, Nout, vt_col0=Nout, rp=pw.Rp):
        ops.PGEMM_CFG[(M, Nout, C, f"vl{pw.Rp}td")] = cfg
        yT = torch.zeros(B, Nout, npad, dtype=torch.bfloat16, device=DEV)
        T = torch.empty(M, pw.Rp, dtype=torch.bfloat16, device=DEV)
        y = ops.conv(dv(x).view(B, 1, N, C), pw, lora_t_out=T, splits=1, vt=yT, vt_col0=0, vt_ld=npad, vt_batch_stride=Nout * npad,
                     vt_dual=True).view(M, Nout)
        close(y, want, rtol=2e-2)
        assert torch.equal(yT[:, :, :N].permute(0, 2, 1).reshape(M, Nout), y), cfg       # the same values, two layouts
        assert not yT[:, :, N:].any()
        close(T[:, :r], x @ A.t(), rtol=2e-2)
    ops.PGEMM_CFG.clear()
    assert labels and all(l[0].startswith("pgemm_") and "_vtd" in l[0] for l in labels), [l[0] for l in labels]


@pytest.mark.parametrize("M,C,nparts", [(1000, 256, 4), (2016, 384, 3), (130, 640, 5)])
def test_geglu_layernorm_folded(ops, labels, M, C, nparts):
    g = torch.Generator().manual_seed(4)
    x = bf(torch.randn(M, C, generator=g) * 1.3 - 0.2)
    w = bf(torch.randn(8 * C, C, generator=g) / math.sqrt(C))
    b = torch.randn(8 * C, generator=g)
    gm, bt = torch.randn(C, generator=g) * 0.3 + 1, torch.randn(C, generator=g) * 0.2
    h = F.layer_norm(x, (C,), gm, bt, 1e-5) @ w.t() + b
    want = h[:, :4 * C] * F.gelu(h[:, 4 * C:])
    pw = ops.pack_linear_ln(w.to(DEV), b.to(DEV), gm.to(DEV), bt.to(DEV), geglu=True)
    lp = _ln_parts(ops, x, nparts)
    for cfg in shapes_for(C, 8 * C, geglu=True):
        ops.PGEMM_CFG[(M, 8 * C, C, "g")] = cfg
        close(ops.linear(dv(x), pw, ln_parts=lp), want, rtol=2e-2)
    # and the plain GEGLU pack (LayerNorm applied by the caller)
    h2 = x @ w.t() + b
    close(ops.linear(dv(x), ops.pack_geglu(w.to(DEV), b.to(DEV))), h2[:, :4 * C] * F.gelu(h2[:, 4 * C:]), rtol=2e-2)
    ops.PGEMM_CFG.clear()
    assert all(l[0].startswith("pgemm_") for l in labels), [l[0] for l in labels]


def test_lora_t_copy_for_the_trainer(ops, labels):
    """lora_t_out: the bf16 T = x A^T the LoRA gradient products consume, identical to what aldm_igemm stores"""
    g = torch.Generator().manual_seed(8)
    M, K, N, r = 1000, 256, 256, 8
    x = dv(torch.randn(M, K, generator=g))
    pw = ops.pack_linear((torch.randn(N, K, generator=g) / 16).to(DEV), torch.randn(N, generator=g).to(DEV))
    A = bf(torch.randn(r, K, generator=g) / r)
    ops.attach_lora(pw, [(0, N, A.to(DEV), (torch.randn(N, r, generator=g) * 0.05).to(DEV), 1.0)])
    res = dv(torch.randn(M, N, generator=g))
    t_new = torch.zeros(M, pw.Rp, dtype=torch.bfloat16, device=DEV)
    t_old = torch.zeros(M, pw.Rp, dtype=torch.bfloat16, device=DEV)
    y_new = ops.linear(x, pw, res=res, lora_t_out=t_new)
    y_old = ops.linear(x, pw, res=res, lora_t_out=t_old, tile=2)                # the convolution kernel
    close(t_new[:, :r], x.float().cpu() @ A.t())
    assert not t_new[:, r:].any()
    assert float((t_new.float() - t_old.float()).abs().max()) <= 2 ** -6 * float(t_old.float().abs().max())
    assert float((y_new.float() - y_old.float()).abs().max()) <= 2 ** -6 * float(y_old.float().abs().max())
    assert labels[0][0].startswith("pgemm_") and labels[1][0].startswith("igemm_")


def test_cold_launches_are_deterministic(ops):
    """Race screen.  Every operand of the kernel reaches LDS asynchronously (LDS-DMA ring, column vectors, LoRA-B rows) behind
    counted waits and barriers; a read that beats its data shows up only when the LDS still holds something else.  So: two different
    problems launched alternately from an idle GPU (each launch finds the other's bytes in LDS), every result compared bit for bit
    with that problem's first result.  (An earlier form of the kernel failed this once in ~100 launches.)"""
    g = torch.Generator().manual_seed(9)
    B, N, C = 2, 1000, 256
    M = B * N
    probs = []
    for k in range(2):
        x = dv(torch.randn(M, C, generator=g) * 1.7 + 0.4)
        pw = ops.pack_linear_ln((torch.randn(3 * C, C, generator=g) / 16).to(DEV), torch.randn(3 * C, generator=g).to(DEV),
                                (torch.randn(C, generator=g) * 0.3 + 1).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV))
        ops.attach_lora(pw, [(i * C, C, (torch.randn(4, C, generator=g) / 4).to(DEV), (torch.randn(C, 4, generator=g) * 0.05).to(DEV), 1.5)
                             for i in range(3)])
        probs.append((x, pw, _ln_parts(ops, x.float().cpu(), 4)))
    key = (M, 3 * C, C, "vl32")
    for cfg in [(2, 32, 1, 4), (1, 32, 1, 4), (2, 32, 6, 4), (2, 64, 3, 4), (1, 64, 1, 4), (1, 32, 1, 8), (1, 32, 6, 8), (1, 64, 3, 8)]:
        ops.PGEMM_CFG[key] = cfg
        first = [None, None]
        for rep in range(60):
            k = rep & 1
            x, pw, lp = probs[k]
            vt = torch.zeros(B, C, N, dtype=torch.bfloat16, device=DEV)
            torch.cuda.synchronize()
            qk = ops.conv(x.view(B, 1, N, C), pw, vt=vt, vt_col0=2 * C, vt_ld=N, vt_batch_stride=C * N, ln_parts=lp)
            got = (qk.clone(), vt)
            if first[k] is None:
                first[k] = got
            else:
                assert torch.equal(got[0], first[k][0]) and torch.equal(got[1], first[k][1]), (cfg, rep)
    ops.PGEMM_CFG.clear()


def test_matches_the_convolution_kernel_it_replaces(ops):
    """same operands through aldm_pgemm and through aldm_igemm (tile forced): equal up to the order of the fp32 sums"""
    g = torch.Generator().manual_seed(5)
    M, K, N = 1000, 256, 256
    x = dv(torch.randn(M, K, generator=g))
    pw = ops.pack_linear((torch.randn(N, K, generator=g) / 16).to(DEV), torch.randn(N, generator=g).to(DEV))
    ops.attach_lora(pw, [(0, N, (torch.randn(4, K, generator=g) / 4).to(DEV), (torch.randn(N, 4, generator=g) * 0.05).to(DEV), 1.0)])
    res = dv(torch.randn(M, N, generator=g))
    a = ops.linear(x, pw, res=res).float()
    b = ops.linear(x, pw, res=res, tile=2).float()
    assert float((a - b).abs().max()) <= 2 ** -6 * float(b.abs().max())


def test_plan_and_refusals(ops):
    import ctypes as C
    from audioldm_with_lora_amd import _lib
    lib = _lib.load()
    a = _lib.PgemmArgs()
    a.M, a.N, a.K = 8000, 768, 256
    assert lib.aldm_pgemm_plan(C.byref(a)) == 0 and a.mi in (1, 2) and a.nt in (32, 64) and (768 // a.nt) % a.tiles_per_range == 0
    a = _lib.PgemmArgs()
    a.M, a.N, a.K = 100, 768, 320
    assert lib.aldm_pgemm_plan(C.byref(a)) != 0                # K the kernel is not built for
    assert lib.aldm_pgemm_supported(256) and lib.aldm_pgemm_supported(640) and not lib.aldm_pgemm_supported(512)
    x = torch.zeros(64, 256, dtype=torch.bfloat16, device=DEV)
    a = _lib.PgemmArgs()
    a.x, a.w, a.out = x.data_ptr(), x.data_ptr(), x.data_ptr()
    a.M, a.N, a.K, a.out_ld = 64, 100, 256, 100               # N not a multiple of 64
    assert lib.aldm_pgemm(C.byref(a), None) != 0
