"""LoRA fine-tune step on the MI355X HIP path (configs 3 / 4).

Replaces the loop body of the reference trainer  [REF script/train/train_audioldm_lora.py:499-565]:
    noisy = scheduler.add_noise(latents, noise, t) ; pred = unet(noisy, t, class_labels=emb)
    loss = mse(pred, noise) ; backward ; AdamW on the LoRA parameters ; polynomial LR
with the base model frozen [REF train:374-376] and LoRA injected by peft-style config [REF train:378-385].

There is no torch autograd here.  A small launch tape records, per forward op, the HIP launches of its backward:
  * dX through convolutions / linears = the SAME implicit-GEMM kernel with transposed (and tap-flipped) weights;
    stride-2 convs use the zero-dilated gather, up-sampler convs the nearest adjoint
  * LoRA sites run forward AND backward as ONE fused GEMM each:   y  = x W^T + (x A^T)(sB)^T        saves T = x A^T
                                                                  dx = dy W  + (dy sB) A            saves U = dy sB
    then  dB = s dy^T T  and  dA = U^T x  are rank-r "reduce over tokens" products scattered with fp32 atomics
    straight into ONE flat gradient buffer -- the buffer RCCL all-reduces and the flat AdamW kernel consumes
  * GroupNorm / LayerNorm / GEGLU / attention backward are dedicated kernels (train.hip, attention_bwd.hip)
All LoRA parameters live in one flat fp32 buffer (the nn.Parameters are views into it); one `aldm_lora_pack`
launch per step refreshes every packed bf16 operand.
"""
import math
import os
import struct
from types import SimpleNamespace

import numpy as np
import torch

from . import dp, ops
from ._lib import ACT_NONE, ACT_SILU
from .lora import LoraLinear
from .unet import UNet2DConditionModel, transformer_gn

BK = 64


# ----------------------------------------------------------------------------------------------
# tape
# ----------------------------------------------------------------------------------------------
class Var:
    __slots__ = ("t", "g", "rg", "gn", "tT", "want_T", "gT", "stats")

    def __init__(self, t, rg=False):
        self.t, self.g, self.rg = t, None, rg
        self.stats = None            # row statistics handed over by the producing GEMM (t_lora_linear rowstats=)
        self.tT = self.want_T = self.gT = None                       # token-major copies (see t_lora_linear)
        self.gn = None               # (channels, groups) when this is a GroupNorm output: its single consumer conv may hand the
                                     # norm's backward its dX as split-K partial tiles (g is then an ops.Deferred)

    @property
    def shape(self):
        return self.t.shape


def acc(v, g):
    if v is None or not v.rg:
        return
    v.g = g if v.g is None else ops.add_bf16(v.g, g.reshape(v.g.shape))


def prior(v, like=None):
    """The gradient v already holds from its other consumers (residual / skip joins), shaped like `like`, or None.
    Kernels that can ADD it (GEMM epilogue `res=`, norm backward `dx_add=`) take it instead of a separate add launch.  They
    write a FRESH tensor: v.g may alias a downstream gradient that a deferred LoRA-gradient job (ops.TnBatch) still reads."""
    if v is None or not v.rg or v.g is None:
        return None
    return v.g if like is None else v.g.view(like.shape)


def put(v, g):
    """Store a gradient that already includes prior(v)."""
    if v is not None and v.rg:
        v.g = g


class Tape:
    def __init__(self, tn=None):
        self.fns = []
        self.tn = tn                 # ops.TnBatch: LoRA-gradient products are deferred to one launch after the backward pass

    def record(self, fn):
        self.fns.append(fn)

    def backward(self):
        for fn in reversed(self.fns):
            fn()
        self.fns.clear()


def _bwd_pack(pw, lo, hi):
    """dX weights from the packed forward weights: Wt[c][(kh', kw', n)] = W[n][KH-1-kh'][KW-1-kw'][c], c in [lo, hi)."""
    key = (lo, hi)
    cache = pw.__dict__.setdefault("_bwd", {})
    if key not in cache:
        n, kh, kw, c = pw.N, pw.KH, pw.KW, pw.Cin
        w = pw.w[:, :kh * kw * c].view(n, kh, kw, c)[:, :, :, lo:hi]
        wt = w.flip(1, 2).permute(3, 1, 2, 0).reshape(hi - lo, kh * kw * n)
        cache[key] = ops.PackedW(ops._pad_k(wt), None, hi - lo, n, kh, kw)
    return cache[key]


def t_view(tape, x, shape):
    y = Var(x.t.view(shape), x.rg)
    if x.rg:
        tape.record(lambda: acc(x, y.g.view(x.t.shape)) if y.g is not None else None)
    return y


def t_conv(tape, x, pw, x2=None, stride=(1, 1), pad=(0, 0), up_size=None, rowbias=None, rowbias_ld=0, res=None, out_f32=False, _gn=None):
    """Frozen conv / linear (no LoRA): forward = ops.conv, backward = dX only."""
    out = ops.conv(x.t, pw, x2=(x2.t if x2 is not None else None), stride=stride, pad=pad, up_size=up_size,
                   rowbias=rowbias, rowbias_ld=rowbias_ld, res=(res.t if res is not None else None), out_f32=out_f32, splits=1 if out_f32 else None,
                   gn=(_gn[0] if _gn else None), gn_keep=bool(_gn))
    if _gn:
        out, _gn[1]["y"] = out                     # t_conv_gn: the GroupNorm of the output came out of the same launches
    y = Var(out)
    y.rg = x.rg or (x2 is not None and x2.rg) or (res is not None and res.rg)
    if not y.rg:
        return y
    c1 = x.t.shape[3]

    def bwd():
        dy = y.g
        if dy is None:
            return
        acc(res, dy)
        bpad = (pw.KH - 1 - pad[0], pw.KW - 1 - pad[1])
        for src, lo, hi in ((x, 0, c1), (x2, c1, pw.Cin)):
            if src is None or not src.rg:
                continue
            bw = _bwd_pack(pw, lo, hi)
            ih, iw = src.t.shape[1], src.t.shape[2]
            if up_size is not None:
                acc(src, ops.upsample_nearest_bwd(ops.conv(dy, bw, pad=bpad), ih, iw))
            elif stride == (2, 2):
                put(src, ops.conv(dy, bw, pad=bpad, in_dilate=2, out_hw=(ih, iw), res=prior(src, src.t)))
            elif src.gn is not None and x2 is None and src.g is None:
                # the input is a GroupNorm output: its backward is the next closure on the tape and sums the partial tiles
                put(src, ops.conv(dy, bw, pad=bpad, defer=src.gn))
            else:
                put(src, ops.conv(dy, bw, pad=bpad, res=prior(src, src.t)))
    tape.record(bwd)
    return y


def t_conv_gn(tape, x, pw, gn, **kw):
    """t_conv then t_groupnorm(*gn) with the two forwards fused (ops.conv gn=: a split-K conv's reduce rides in the norm);
    returns (conv output, its GroupNorm) -- the backward is the unfused pair's."""
    box = {}
    y = t_conv(tape, x, pw, _gn=(gn, box), **kw)
    return y, t_groupnorm(tape, y, *gn, fwd=box["y"])


def t_groupnorm(tape, x, gamma, beta, groups, eps, act, x2=None, fwd=None):
    y = Var(fwd if fwd is not None else ops.groupnorm(x.t, gamma, beta, groups, eps, act, x2=(x2.t if x2 is not None else None)))
    y.rg = x.rg or (x2 is not None and x2.rg)
    if y.rg:
        y.gn = (y.t.shape[3], groups)

        def bwd():
            if y.g is None:
                return
            dx, dx2 = ops.groupnorm_bwd(x.t, y.g, gamma, beta, groups, eps, act, x2=(x2.t if x2 is not None else None),
                                        need_dx2=(x2 is not None and x2.rg), dx_add=prior(x), dx2_add=prior(x2))
            put(x, dx)
            put(x2, dx2)
        tape.record(bwd)
    return y


def t_layernorm(tape, x, gamma, beta):
    y = Var(ops.layernorm(x.t, gamma, beta), x.rg)
    if x.rg:
        tape.record(lambda: put(x, ops.layernorm_bwd(x.t, y.g, gamma, dx_add=prior(x))) if y.g is not None else None)
    return y


def t_layernorm_folded(tape, x, gamma):
    """LayerNorm whose FORWARD is folded into the consuming GEMM (weights packed with pack_linear_ln, row statistics handed over by
    the producer of x: no launch here).  The returned Var stands for LN(x) on the tape -- its tensor is x itself, which the folded
    GEMM reads -- and the backward is the ordinary LayerNorm backward (it recomputes the statistics from x).  Only valid where
    nothing but that GEMM consumes LN(x): norm3 -> GEGLU (no LoRA gradient reads the normalised activations there)."""
    y = Var(x.t, x.rg)
    if x.rg:
        tape.record(lambda: put(x, ops.layernorm_bwd(x.t, y.g, gamma, dx_add=prior(x))) if y.g is not None else None)
    return y


# ----------------------------------------------------------------------------------------------
# LoRA sites: one per fused projection GEMM (q|k|v together, out-proj alone)
# ----------------------------------------------------------------------------------------------
class LoraSite:
    """Packed operands + gradient scatter tables of one LoRA-bearing GEMM."""

    def __init__(self, weight, bias, parts, flat, dev):
        # parts: list of (row0, nrows, A param, B param, scaling)
        self.N, self.K = weight.shape
        self.parts = parts
        rtot = sum(p[2].shape[0] for p in parts)
        self.Rp = 32 if rtot <= 32 else 64
        assert rtot <= 64, "combined LoRA rank per GEMM must be <= 64"
        npad = (self.N + BK - 1) // BK * BK
        self.fwd = ops.pack_linear(weight, bias)
        self.fwd.lora_a = torch.zeros(self.Rp, self.fwd.Kpad, dtype=torch.bfloat16, device=dev)
        self.fwd.lora_b = torch.zeros(self.N, self.Rp, dtype=torch.bfloat16, device=dev)
        self.fwd.Rp = self.Rp
        self.bwd = ops.PackedW(ops._pad_k(weight.detach().t().contiguous()), None, self.K, self.N)
        self.bwd.lora_a = torch.zeros(self.Rp, npad, dtype=torch.bfloat16, device=dev)
        self.bwd.lora_b = torch.zeros(self.K, self.Rp, dtype=torch.bfloat16, device=dev)
        self.bwd.Rp = self.Rp
        self.fwd.ranks_used = self.bwd.ranks_used = rtot     # (<= 32: the K = 256 / 384 / 640 launches may run on aldm_pgemm)
        self.jobs = []
        rows_db = [(0, 0, 0, 0, 0.0)] * self.Rp
        rows_da = [(0, 0, 0, 0, 0.0)] * self.Rp
        col = 0
        for row0, nrows, A, Bm, s in parts:
            r, k = A.shape
            oa, ob = flat.offset_of(A), flat.offset_of(Bm)
            pa, pb = flat.params.data_ptr() + 4 * oa, flat.params.data_ptr() + 4 * ob
            ga, gb = flat.grads.data_ptr() + 4 * oa, flat.grads.data_ptr() + 4 * ob
            self.jobs += [
                (pa, self.fwd.lora_a.data_ptr() + 2 * col * self.fwd.Kpad, r, k, k, self.fwd.Kpad, 0, 1.0),
                (pb, self.fwd.lora_b.data_ptr() + 2 * (row0 * self.Rp + col), nrows, r, r, self.Rp, 0, s),
                (pb, self.bwd.lora_a.data_ptr() + 2 * (col * npad + row0), nrows, r, r, npad, 1, s),
                (pa, self.bwd.lora_b.data_ptr() + 2 * col, r, k, k, self.Rp, 1, 1.0),
            ]
            for j in range(r):
                rows_db[col + j] = (gb + 4 * j, row0, row0 + nrows, r, s)        # dB[n][j] = s * sum_m dy[m][n] T[m][col+j]
                rows_da[col + j] = (ga + 4 * j * k, 0, k, 1, 1.0)                # dA[j][k] = sum_m U[m][col+j] x[m][k]
            col += r
        pack = lambda rows: torch.frombuffer(bytearray(b"".join(struct.pack("<qiiif", *r) for r in rows)), dtype=torch.uint8).to(dev)
        self.rows_db, self.rows_da = pack(rows_db), pack(rows_da)


def _tok_major_buf(B, N, Cc, dev):
    npad = (N + 7) // 8 * 8
    mk = torch.empty if npad == N else torch.zeros            # the GEMM writes tokens < N only: keep the padding zero
    return mk(B, Cc, npad, dtype=torch.bfloat16, device=dev)


def t_lora_linear(tape, x, site, res=None, tok=None, rowstats=False):
    """x [M, K] -> y [M, N] with the LoRA side channel; records dX + dA/dB.
    tok = (B, N): the GEMM ALSO stores y token-major ([B, N_out, Npad], aldm_igemm vt_dual) -- returned as y.tT; the flash
    kernels read q | k | v that way, so the separate transpose launch disappears.  A Var x with x.want_T = (B, N) asks the same
    of its GRADIENT (the attention backward reads dO token-major): the dX launch then leaves it in x.gT."""
    M = x.t.shape[0]
    T = torch.empty(M, site.Rp, dtype=torch.bfloat16, device=x.t.device)
    if tok is not None and res is None:
        Bq, Nq = tok
        yT = _tok_major_buf(Bq, Nq, site.N, x.t.device)
        out = ops.conv(x.t.view(Bq, 1, Nq, site.K), site.fwd, lora_t_out=T, splits=1, vt=yT, vt_col0=0, vt_ld=yT.shape[2],
                       vt_batch_stride=site.N * yT.shape[2], vt_dual=True).view(M, site.N)
        y = Var(out, True)
        y.tT = yT
    else:
        out = ops.linear(x.t, site.fwd, res=(res.t if res is not None else None), lora_t_out=T, splits=1, rowstats=rowstats)
        st = None
        if rowstats:
            out, st = out
        y = Var(out, True)
        y.stats = st                 # per-row partial (sum, sum of squares) of the stored values: a LayerNorm-folded consumer's ln_parts

    def bwd():
        dy = y.g
        if dy is None:
            return
        acc(res, dy)
        U = torch.empty(M, site.Rp, dtype=torch.bfloat16, device=dy.device)
        g0 = prior(x, x.t)
        want = getattr(x, "want_T", None)
        if want is not None and g0 is None:                                  # dx row-major AND token-major from one launch
            Bq, Nq = want
            gT = _tok_major_buf(Bq, Nq, site.K, dy.device)
            put(x, ops.conv(dy.view(Bq, 1, Nq, site.N), site.bwd, lora_t_out=U, splits=1, vt=gT, vt_col0=0, vt_ld=gT.shape[2],
                            vt_batch_stride=site.K * gT.shape[2], vt_dual=True).view(M, site.K))
            x.gT = gT
        else:
            put(x, ops.linear(dy, site.bwd, lora_t_out=U, splits=1, res=g0)) # dx = dy W + (dy sB) A (+ prior) ; U = dy sB
        if tape.tn is not None:
            tape.tn.add(T, dy, site.rows_db, site.N)
            tape.tn.add(U, x.t, site.rows_da, site.K)
        else:
            ops.tn_small(T, dy, site.rows_db, Qc=site.N)
            ops.tn_small(U, x.t, site.rows_da, Qc=site.K)
    tape.record(bwd)
    return y


def _lin_parts(mod, row0):
    if isinstance(mod, LoraLinear):
        return (row0, mod.base_layer.out_features, mod.lora_A["default"].weight, mod.lora_B["default"].weight, mod.scaling)
    return None


def _base(mod):
    return mod.base_layer if isinstance(mod, LoraLinear) else mod


# ----------------------------------------------------------------------------------------------
class FlatLora:
    """All LoRA parameters as ONE fp32 buffer (+ grads, Adam moments); the module parameters become views."""

    def __init__(self, model, device):
        named = [(n, p) for n, p in model.named_parameters() if "lora_" in n]
        self.names = [n for n, _ in named]
        total = sum(p.numel() for _, p in named)
        self.n = total
        self.params = torch.zeros(total, dtype=torch.float32, device=device)
        self.grads = torch.zeros(total + 1, dtype=torch.float32, device=device)     # last slot: loss (rides the all-reduce)
        self.m = torch.zeros(total, dtype=torch.float32, device=device)
        self.v = torch.zeros(total, dtype=torch.float32, device=device)
        self._off = {}
        self._plist = []
        self.on_change = None             # set by the owning trainer: called after any optimiser update of the buffer
        off = 0
        for n, p in named:
            k = p.numel()
            self.params[off:off + k].copy_(p.detach().reshape(-1).to(device, torch.float32))
            p.data = self.params[off:off + k].view(p.shape)
            p.grad = self.grads[off:off + k].view(p.shape)
            p.requires_grad_(True)
            p._aldm_flat = (self, off)    # optim.AdamW finds the flat buffer through its parameters
            self._off[id(p)] = off
            self._plist.append((p, off, k))
            off += k

    def offset_of(self, p):
        return self._off[id(p)]

    def bind_grads(self):
        """(Re-)attach every parameter's .grad as a view of the flat gradient buffer (optimizer.zero_grad() sets them to None)."""
        for p, off, k in self._plist:
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * off:
                p.grad = self.grads[off:off + k].view(p.shape)

    def intact(self):
        """False once a parameter no longer aliases the flat buffer (module.to(), a state load that re-created the tensors)."""
        base = self.params.data_ptr()
        return all(p.data_ptr() == base + 4 * off for p, off, _ in self._plist)

    def owner_changed(self):
        if self.on_change is not None:
            self.on_change()


class _UNetTrainFn(torch.autograd.Function):
    """`unet(noisy, t, class_labels=emb)[0]` for a LoRA-wrapped UNet in training mode  [REF script/train/train_audioldm_lora.py:539-546]:
    forward = the taped HIP launch sequence, backward = the tape run in reverse (dX through the frozen base, dA / dB scattered
    into the flat gradient buffer).  The LoRA parameters are inputs of the Function only so that autograd schedules it; their
    gradients are WRITTEN by the kernels into the flat buffer and `p.grad` is bound to views of it -- nothing is returned for
    them, so autograd makes no copies and `accelerator.backward(loss)` / `optimizer.step()` [REF train:557-565] see one buffer."""

    @staticmethod
    def forward(ctx, trainer, sample, timestep, class_labels, *lora_params):
        ctx.trainer = trainer
        ctx.nparams = len(lora_params)
        return trainer._taped_forward(sample, timestep, class_labels)

    @staticmethod
    def backward(ctx, grad_out):
        ctx.trainer._taped_backward(grad_out)
        return (None, None, None, None) + (None,) * ctx.nparams


def trainer_of(unet, create=True):
    """The training engine attached to a UNet (created on first use: flattens the LoRA parameters, builds the fused sites)."""
    tr = getattr(unet, "_trainer", None)
    old = None
    if tr is not None and not tr.flat.intact():
        old, tr = tr, None                                   # the LoRA parameters were moved / re-created: rebuild
    if tr is None and create:
        if old is not None:
            tr = LoraTrainer(unet, old.scheduler, lr=old.lr0, betas=old.betas, weight_decay=old.wd, eps=old.eps,
                             max_train_steps=old.max_train_steps, lr_end=old.lr_end, power=old.power, use_graph=old.use_graph)
            if tr.flat.names != old.flat.names or tr.flat.n != old.flat.n:
                raise ops._lib.AldmError("trainer_of: the set of LoRA parameters changed under a live training engine -- its AdamW "
                                         "moments cannot be carried over; build a new LoraTrainer / optimizer explicitly")
            # the moments (and the step count that bias-corrects them) belong to the parameters, not to the buffer they lived in
            tr.flat.m.copy_(old.flat.m.to(tr.flat.m.device))
            tr.flat.v.copy_(old.flat.v.to(tr.flat.v.device))
            tr.step_count = old.step_count
        else:
            tr = LoraTrainer(unet, None, use_graph=False)
    elif tr is not None:
        tr._ensure_fresh()
    return tr


FOLD_NORM3 = os.environ.get("ALDM_NO_FOLD_NORM3") is None             # training forward: norm3 folded into the GEGLU GEMM (A/B aid)
SIDE_STREAM_H2D = os.environ.get("ALDM_NO_SIDE_STREAM_H2D") is None      # step_from_batch: batch copies on a side stream (A/B aid)


class LoraTrainer:
    """One process per GPU; `world`/`rank` follow torch.distributed when it is initialised (RCCL over xGMI)."""

    def __init__(self, unet: UNet2DConditionModel, scheduler=None, lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-5, eps=1e-8,
                 max_train_steps=97000, lr_end=1e-7, power=1.0, device="cuda", use_graph=True):
        self.unet, self.scheduler = unet, scheduler
        self.dev = torch.device(device)
        self.lr0, self.betas, self.wd, self.eps = lr, betas, weight_decay, eps
        self.max_train_steps, self.lr_end, self.power = max_train_steps, lr_end, power
        self.step_count = 0
        self.dist = torch.distributed if (torch.distributed.is_available() and torch.distributed.is_initialized()) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        if not any("lora_" in n for n, _ in unet.named_parameters()):
            raise ops._lib.AldmError("LoraTrainer: the UNet carries no LoRA parameters (call get_peft_model first)")
        self.flat = FlatLora(unet, self.dev)
        self.flat.on_change = unet.invalidate_packed         # an optimiser update makes the UNet's inference plan stale
        self.use_graph, self.graph, self._static, self._eager_steps = use_graph, None, None, 0
        self._graph_tab, self._body_graphs, self._body_eager = None, {}, {}
        self.ac_dev = scheduler.alphas_cumprod.to(self.dev, torch.float32) if scheduler is not None else None
        self._tape = None
        self._build_sites()
        self.weights_version = unet._weights_version
        unet.__dict__["_trainer"] = self                     # plain attribute (not a submodule): unet(...) in training mode finds it
        self.tnb = ops.TnBatch(4 * len(self.sites) + 8, self.dev)       # <= 2 sites per attention, 2 products per site
        dp.broadcast_(self.flat.params, src=0)                  # DDP's initial parameter broadcast (C3), LoRA buffer only

    # ---- site construction ----
    def _build_sites(self):
        u = self.unet
        P = u.plan()                                             # frozen packed weights (GN params, convs, ...)
        self.P = P
        jobs = []
        self.sites = {}

        def attn_sites(a):
            wq, wk, wv, wo = _base(a.to_q), _base(a.to_k), _base(a.to_v), _base(a.to_out[0])
            c = wq.weight.shape[0]
            bias = torch.cat([wq.bias, wk.bias, wv.bias]) if wq.bias is not None else None
            parts = [p for p in (_lin_parts(a.to_q, 0), _lin_parts(a.to_k, c), _lin_parts(a.to_v, 2 * c)) if p]
            qkv = LoraSite(torch.cat([wq.weight, wk.weight, wv.weight]), bias, parts, self.flat, self.dev) if parts else None
            po = _lin_parts(a.to_out[0], 0)
            out = LoraSite(wo.weight, wo.bias, [po], self.flat, self.dev) if po else None
            for s in (qkv, out):
                if s:
                    jobs.extend(s.jobs)
            return qkv, out

        blocks = []
        for blk in list(u.down_blocks) + [u.mid_block] + list(u.up_blocks):
            if getattr(blk, "has_attn", True) and hasattr(blk, "attentions"):
                blocks.extend(blk.attentions)
        for t in blocks:
            tb = t.transformer_blocks[0]
            self.sites[id(tb.attn1)] = attn_sites(tb.attn1)
            self.sites[id(tb.attn2)] = attn_sites(tb.attn2)
        rec = b"".join(struct.pack("<qqiiiiif", *j) for j in jobs)
        self.njobs = len(jobs)
        self.jobs_dev = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(self.dev)

    def repack(self):
        ops.lora_pack(self.jobs_dev, self.njobs)

    def _ensure_fresh(self):
        """The FROZEN weights may have changed in place under a live engine (load_state_dict of the base, `.to()`): the LoRA views are
        intact, but the packed copies of the base weights (plan + LoraSite operands) are stale, and every captured graph holds pointers
        into the operands `_build_sites` is about to free.  Rebuild the sites and drop EVERY graph (step's and step_from_batch's) and
        their warm-up counters.  Called at the top of every entry point that launches, and by trainer_of."""
        if self.weights_version == self.unet._weights_version:
            return
        self._build_sites()
        self.graph, self._static, self._eager_steps = None, None, 0
        self._graph_tab = None
        self._body_graphs, self._body_eager = {}, {}
        self.tnb.owner = None
        self.weights_version = self.unet._weights_version

    # ---- forward with tape ----
    def _attention(self, tape, tblk_attn, Pa, hn, h_res, B, N, rowstats=False):
        qkv_site, out_site = self.sites[id(tblk_attn)]
        C, H, d = Pa.c, Pa.heads, Pa.d
        if qkv_site is not None:
            qkv = t_lora_linear(tape, hn, qkv_site, tok=(B, N))
        else:
            if "qkv_plain" not in Pa.__dict__:
                wq, wk, wv = _base(tblk_attn.to_q), _base(tblk_attn.to_k), _base(tblk_attn.to_v)
                bias = torch.cat([wq.bias, wk.bias, wv.bias]) if wq.bias is not None else None
                Pa.qkv_plain = ops.pack_linear(torch.cat([wq.weight, wk.weight, wv.weight]), bias)
            qkv = t_conv(tape, t_view(tape, hn, (1, 1, B * N, C)), Pa.qkv_plain)
            qkv = t_view(tape, qkv, (B * N, 3 * C))
        qkvT = qkv.tT if qkv.tT is not None else ops.transpose_tokens(qkv.t, B, N, 3 * C)
        o_t, lse = ops.attention_train(qkv.t, qkvT, B, N, H, d)
        o = Var(o_t, True)
        o.want_T = (B, N)                                   # the out-projection's dX launch leaves dO token-major in o.gT

        def bwd():
            if o.g is not None:
                acc(qkv, ops.attention_bwd(qkv.t, qkvT, o.g, o.t, lse, B, N, H, d, dOT=o.gT))
        if qkv.rg:
            tape.record(bwd)
        o.rg = qkv.rg
        if out_site is not None:
            return t_lora_linear(tape, o, out_site, res=h_res, rowstats=rowstats)
        y = t_conv(tape, t_view(tape, o, (1, 1, B * N, C)), Pa.out, res=t_view(tape, h_res, (1, 1, B * N, C)))
        return t_view(tape, y, (B * N, C))

    def _transformer(self, tape, tmod, Pt, x, xn=None):
        B, H, W, C = x.shape
        N = H * W
        tb = tmod.transformer_blocks[0]
        h = xn if xn is not None else t_groupnorm(tape, x, *transformer_gn(Pt))
        h = t_view(tape, t_conv(tape, h, Pt.proj_in), (B * N, C))
        h = self._attention(tape, tb.attn1, Pt.attn1, t_layernorm(tape, h, *Pt.ln[0]), h, B, N)
        # norm3 -> GEGLU: nothing but the GEGLU projection consumes LN(h) (no LoRA site there), so its FORWARD is folded into that GEMM as
        # in inference (Pt.ff1 = W diag(gamma) with the row statistics handed over by the out-projection); the backward is unchanged
        fold3 = FOLD_NORM3 and C % 64 == 0 and getattr(Pt.ff1, "ln_s", None) is not None and self.sites[id(tb.attn2)][1] is not None
        h = self._attention(tape, tb.attn2, Pt.attn2, t_layernorm(tape, h, *Pt.ln[1]), h, B, N, rowstats=fold3)
        st3 = getattr(h, "stats", None) if fold3 else None
        if st3 is not None and st3.shape[1] > 16:
            st3 = None                                          # (more partials per row than the consumer takes: keep the launch)
        n3 = t_layernorm_folded(tape, h, Pt.ln[2][0]) if st3 is not None else t_layernorm(tape, h, *Pt.ln[2])
        if "ff1_plain" not in Pt.__dict__:                    # LayerNorm NOT folded (the backward needs LN(h))
            gp = ops.pack_geglu(tb.ff.net[0].proj.weight, tb.ff.net[0].proj.bias)
            Pt.ff1_plain = ops.PackedW(gp.w, gp.bias, gp.N, gp.Cin)           # same packed rows, plain-linear view (for dX)
            Pt.ff1_geglu = gp
        ff1 = Pt.ff1_plain
        if C % 64 == 0:
            # one GEGLU GEMM writes g = value * gelu(gate) AND keeps the pre-activation projection for the backward
            M = B * N
            hp_t = torch.empty(M, 8 * C, dtype=torch.bfloat16, device=n3.t.device)
            if st3 is not None:
                g = Var(ops.conv(n3.t.view(1, 1, M, C), Pt.ff1, out2=hp_t, splits=1, ln_parts=st3).view(M, 4 * C), n3.rg)
            else:
                g = Var(ops.conv(n3.t.view(1, 1, M, C), Pt.ff1_geglu, out2=hp_t, splits=1).view(M, 4 * C), n3.rg)
            if n3.rg:
                def bwd_ff1():
                    if g.g is None:
                        return
                    dhp = ops.geglu_bwd(hp_t, g.g)
                    put(n3, ops.conv(dhp.view(1, 1, M, 8 * C), _bwd_pack(ff1, 0, C), res=prior(n3, n3.t.view(1, 1, M, C))).view(M, C))
                tape.record(bwd_ff1)
        else:                                                 # narrow test configurations: un-fused projection + GEGLU kernels
            hp = t_view(tape, t_conv(tape, t_view(tape, n3, (1, 1, B * N, C)), ff1), (B * N, 8 * C))
            g = Var(ops.geglu_fwd(hp.t), hp.rg)
            if hp.rg:
                tape.record(lambda: acc(hp, ops.geglu_bwd(hp.t, g.g)) if g.g is not None else None)
        if Pt.ff2_proj is not None:           # ff2 and proj_out as one GEMM over the virtual concat [g | h] (see unet._merge_ff2_proj_out);
            # its dX splits back into dg = dy (Wp W2) and dh = dy Wp -- the same two backward GEMMs as before
            return t_conv(tape, t_view(tape, g, (B, H, W, 4 * C)), Pt.ff2_proj, x2=t_view(tape, h, (B, H, W, C)), res=x)
        h4 = t_view(tape, h, (1, 1, B * N, C))
        h = t_conv(tape, t_view(tape, g, (1, 1, B * N, 4 * C)), Pt.ff2, res=h4)
        return t_conv(tape, t_view(tape, h, (B, H, W, C)), Pt.proj_out, res=x)

    @staticmethod
    def _resnet(tape, Pr, x, x2, rowbias, ld, next_gn=None):
        """next_gn: GroupNorm of the Transformer2DModel behind the block -> returns (output, its norm) as unet.run_resnet"""
        # the shortcut is recorded FIRST: on the way back every conv's dX is then followed directly by the backward of the
        # GroupNorm that produced its input (conv2 -> norm2 -> conv1 -> norm1 -> shortcut), which lets a split-K dX conv leave
        # its reduce to that norm (t_conv / ops.groupnorm_bwd with an ops.Deferred)
        xs = t_conv(tape, x, Pr.shortcut, x2=x2) if Pr.shortcut is not None else x
        h = t_groupnorm(tape, x, Pr.g1, Pr.b1, Pr.groups, Pr.eps, ACT_SILU, x2=x2)
        _, h = t_conv_gn(tape, h, Pr.conv1, (Pr.g2, Pr.b2, Pr.groups, Pr.eps, ACT_SILU), pad=(1, 1),
                         rowbias=rowbias[:, Pr.temb_off:], rowbias_ld=ld)
        if next_gn is not None:
            return t_conv_gn(tape, h, Pr.conv2, next_gn, pad=(1, 1), res=xs)
        return t_conv(tape, h, Pr.conv2, pad=(1, 1), res=xs)

    def forward(self, tape, x_in, t_dev, cls_bf16):
        ops.drop_pending()
        u, P = self.unet, self.P
        cfg = u.cfg
        b, H, W, _ = x_in.shape
        boc = cfg["block_out_channels"]
        ted = boc[0] * 4
        groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        factor = 2 ** u.num_upsamplers
        fus = (H % factor != 0) or (W % factor != 0)
        temb = ops.timestep_embedding(t_dev, b, boc[0])
        e1 = ops.linear(temb, P.te1, out_act=ACT_SILU)
        semb = torch.empty(b, 2 * ted, dtype=torch.bfloat16, device=x_in.device)
        ops.linear(e1, P.te2, out_act=ACT_SILU, out=semb, out_ld=2 * ted)
        ops.linear(cls_bf16, P.cls, out_act=ACT_SILU, out=semb[:, ted:], out_ld=2 * ted)
        rowbias = ops.linear(semb, P.temb_all, out_f32=True)
        ld = P.temb_total

        h = Var(ops.conv(x_in, P.conv_in, pad=(1, 1)))
        skips = [h]
        for blk, Pb in zip(u.down_blocks, P.down):
            for i, r in enumerate(Pb.resnets):
                if Pb.attns is not None:
                    h, hn = self._resnet(tape, r, h, None, rowbias, ld, next_gn=transformer_gn(Pb.attns[i]))
                    h = self._transformer(tape, blk.attentions[i], Pb.attns[i], h, xn=hn)
                else:
                    h = self._resnet(tape, r, h, None, rowbias, ld)
                skips.append(h)
            if Pb.down is not None:
                h = t_conv(tape, h, Pb.down, stride=(2, 2), pad=(1, 1))
                skips.append(h)
        h, hn = self._resnet(tape, P.mid.resnets[0], h, None, rowbias, ld, next_gn=transformer_gn(P.mid.attns[0]))
        h = self._transformer(tape, u.mid_block.attentions[0], P.mid.attns[0], h, xn=hn)
        h = self._resnet(tape, P.mid.resnets[1], h, None, rowbias, ld)
        for blk, Pb in zip(u.up_blocks, P.up):
            for i, r in enumerate(Pb.resnets):
                if Pb.attns is not None:
                    h, hn = self._resnet(tape, r, h, skips.pop(), rowbias, ld, next_gn=transformer_gn(Pb.attns[i]))
                    h = self._transformer(tape, blk.attentions[i], Pb.attns[i], h, xn=hn)
                else:
                    h = self._resnet(tape, r, h, skips.pop(), rowbias, ld)
            if Pb.up is not None:
                size = (skips[-1].shape[1], skips[-1].shape[2]) if fus else (h.shape[1] * 2, h.shape[2] * 2)
                h = t_conv(tape, h, Pb.up, pad=(1, 1), up_size=size)
        h = t_groupnorm(tape, h, P.gn_out[0], P.gn_out[1], groups, eps, ACT_SILU)
        return t_conv(tape, h, P.conv_out, pad=(1, 1), out_f32=True)

    # ---- autograd-shaped boundary (the reference's own loop body drives these through unet(...) / loss.backward()) ----
    def _taped_forward(self, sample, timestep, class_labels):
        self._ensure_fresh()
        f = self.flat
        b = sample.shape[0]
        self.repack()
        f.grads.zero_()
        t = timestep.to(device=self.dev, dtype=torch.float32).reshape(-1)
        if t.numel() == 1 and b > 1:
            t = t.expand(b)
        x_in = ops.nchw_to_nhwc(sample.detach().to(self.dev, torch.float32).contiguous())
        cls = ops.f32_to_bf16(class_labels.detach().to(self.dev, torch.float32).contiguous())
        tape = Tape(self.tnb)
        pred = self.forward(tape, x_in, t.contiguous(), cls)
        self._tape = (tape, pred)
        return ops.nhwc_to_nchw_f32(pred.t).to(sample.dtype)

    def _taped_backward(self, grad_out):
        if self._tape is None:
            raise ops._lib.AldmError("backward called twice on one UNet forward (the launch tape is consumed by the first call)")
        tape, pred = self._tape
        self._tape = None
        pred.g = ops.nchw_to_nhwc(grad_out.detach().to(self.dev, torch.float32).contiguous())     # bf16 activation-gradient
        tape.backward()
        self.tnb.launch()
        self.flat.bind_grads()

    def autograd_forward(self, sample, timestep, class_labels):
        params = [p for p, _, _ in self.flat._plist]
        return _UNetTrainFn.apply(self, sample, timestep, class_labels, *params)

    def allreduce_grads_(self):
        """DDP's gradient all-reduce (C1) for the autograd-shaped path: ONE collective over the flat buffer, then 1 / world."""
        if self.dist and self.world > 1:
            self.dist.all_reduce(self.flat.grads)
            self.flat.grads.mul_(1.0 / self.world)

    # ---- schedule / step ----
    def lr(self, step):
        """diffusers get_scheduler("polynomial") with 0 warm-up [REF train:438-443]."""
        if step > self.max_train_steps:
            return self.lr_end
        return (self.lr0 - self.lr_end) * (1 - step / self.max_train_steps) ** self.power + self.lr_end

    def _fwd_bwd(self, latents, noise, timesteps, prompt_embeds):
        """Device-only launch sequence (capturable): repack LoRA -> add_noise -> forward tape -> MSE -> backward."""
        f = self.flat
        self.repack()
        f.grads.zero_()
        if self.ac_dev is None:
            raise ops._lib.AldmError("LoraTrainer.step needs the noise scheduler (pass scheduler= to the constructor)")
        noisy = ops.add_noise_t(latents, noise, self.ac_dev, timesteps)      # coefficients looked up on the device: one launch
        x_in = ops.nchw_to_nhwc(noisy)
        tgt = ops.nchw_to_nhwc(noise, out_f32=True)
        t_dev = timesteps.to(torch.float32)
        cls = ops.f32_to_bf16(prompt_embeds)
        tape = Tape(self.tnb)
        pred = self.forward(tape, x_in, t_dev, cls)
        pred.g = ops.mse_grad(pred.t, tgt, f.grads[f.n:])
        tape.backward()
        self.tnb.launch()                                        # every dA / dB of the step: one launch (two if Rp differs)

    def _to_dev(self, latents, noise, timesteps, prompt_embeds):
        return (latents.to(self.dev, torch.float32).contiguous(), noise.to(self.dev, torch.float32).contiguous(),
                timesteps.to(self.dev, torch.int64).contiguous(), prompt_embeds.to(self.dev, torch.float32).contiguous())

    def loss_and_grads(self, latents, noise, timesteps, prompt_embeds):
        """Fills the flat gradient buffer (this rank's batch-mean loss in the last slot); returns that loss slot.
        After two eager warm-up steps the whole launch sequence (~1000 kernels) is captured once into a hipGraph and
        replayed from static input buffers -- the eager step is host-launch-bound."""
        self._ensure_fresh()
        f = self.flat
        args = self._to_dev(latents, noise, timesteps, prompt_embeds)
        if not self.use_graph:
            self._fwd_bwd(*args)
            return f.grads[f.n:]
        if self.graph is not None and all(a.shape == b.shape for a, b in zip(args, self._static)):
            for dst, src in zip(self._static, args):
                dst.copy_(src)
            self.tnb.restore(self._graph_tab)                  # the job table of THIS graph's addresses, if anything overwrote it
            self.graph.replay()
            return f.grads[f.n:]
        self._eager_steps += 1
        if self._eager_steps <= 2:
            self._fwd_bwd(*args)
            return f.grads[f.n:]
        self._static = tuple(a.clone() for a in args)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):   # RCCL's watchdog thread must not void the capture
            self._fwd_bwd(*self._static)
        self._graph_tab = self.tnb.snapshot()                    # job table of the captured addresses (H2D copies cannot be captured)
        self.tnb.restore(self._graph_tab)
        self.graph.replay()
        return f.grads[f.n:]

    def step_from_batch(self, vae, text_encoder, batch, noise, timesteps, sample_noise):
        """The reference's WHOLE loop body for one collate_fn batch [REF script/train/train_audioldm_lora.py:495-565]:
            latents = vae.encode(batch["log_mel_spec"]).latent_dist.sample() * vae.config.scaling_factor
            prompt_embeds = normalize(text_encoder(input_ids, attention_mask).text_embeds)
            noisy = add_noise(latents, noise, timesteps);  pred = unet(noisy, t, class_labels=prompt_embeds);  mse;  backward
        as ONE captured hipGraph per (batch shape, caption-length bucket), then the flat all-reduce and AdamW as in step().
        sample_noise [B, 8, H/4, 16] is the N(0, 1) draw of latent_dist.sample() (passed in so that the graph has no RNG state)."""
        self._ensure_fresh()
        dev = self.dev
        ids, mask = batch["input_ids"].squeeze(1), batch["attention_mask"].squeeze(1)
        lens = text_encoder._lengths(ids, mask)                  # host-side validation of the padding mask (CPU tensors)
        Le = min(ids.shape[1], (int(lens.max()) + 63) // 64 * 64)   # 64-token buckets: few distinct graphs
        # Host -> device on a SIDE stream: a pageable `.to(device)` blocks the host until the copy has run, and on the step's own stream
        # that means until the PREVIOUS step's graph has finished -- the host-side work of a step (~0.3 ms: slicing, six copies, the
        # replay call) then sits between two graphs instead of under the first.  The side stream is idle, so the host only waits for
        # the copies themselves; the step's stream picks the staged tensors up behind an event.
        if SIDE_STREAM_H2D:
            if getattr(self, "_h2d_stream", None) is None:
                self._h2d_stream = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            with torch.cuda.stream(self._h2d_stream):
                args = (batch["log_mel_spec"].to(dev, torch.float32).contiguous(), ids[:, :Le].to(dev, torch.int64).contiguous(),
                        lens.to(dev), sample_noise.to(dev, torch.float32).contiguous(), noise.to(dev, torch.float32).contiguous(),
                        timesteps.to(dev, torch.int64).contiguous())
            main.wait_stream(self._h2d_stream)
            for a in args:
                a.record_stream(main)
        else:
            args = (batch["log_mel_spec"].to(dev, torch.float32).contiguous(), ids[:, :Le].to(dev, torch.int64).contiguous(),
                    lens.to(dev), sample_noise.to(dev, torch.float32).contiguous(), noise.to(dev, torch.float32).contiguous(),
                    timesteps.to(dev, torch.int64).contiguous())
        sf = float(vae.config.scaling_factor)

        def body(mel, ids_d, kv_len, eps, nz, ts):
            mom = ops.nhwc_to_nchw_f32(vae.encode_nhwc(ops.nchw_to_nhwc(mel)))
            lat = ops.gaussian_sample(mom, eps) * sf
            emb = torch.nn.functional.normalize(text_encoder.forward_device(ids_d, kv_len)[0], dim=-1)
            self._fwd_bwd(lat, nz, ts, emb)

        f = self.flat
        key = tuple(tuple(a.shape) for a in args)
        if not self.use_graph:
            body(*args)
        else:
            ent = self._body_graphs.get(key)
            if ent is None:
                n = self._body_eager.get(key, 0) + 1
                self._body_eager[key] = n
                if n <= 2:
                    body(*args)                                # two eager warm-up steps (code objects, workspace sizes)
                else:
                    static = tuple(a.clone() for a in args)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        body(*static)
                    # every graph's aldm_tn_batched launch reads the trainer's ONE device job table: each graph keeps the table of ITS
                    # captured addresses and puts it back before a replay whenever another graph, a capture or an eager step overwrote it
                    tab = self.tnb.snapshot()
                    self._body_graphs[key] = (g, static, tab)
                    self.tnb.restore(tab)
                    g.replay()
            else:
                g, static, tab = ent
                for dst, src in zip(static, args):
                    dst.copy_(src)
                self.tnb.restore(tab)
                g.replay()
        return self._apply_update(f.grads[f.n:])

    def step(self, latents, noise, timesteps, prompt_embeds):
        """One optimisation step; returns the (all-rank mean) loss as a device tensor."""
        return self._apply_update(self.loss_and_grads(latents, noise, timesteps, prompt_embeds))

    def _apply_update(self, loss):
        f = self.flat
        if self.dist and self.world > 1:
            self.dist.all_reduce(f.grads)                         # ONE flat all-reduce: every LoRA grad + the loss slot
        self.step_count += 1
        ops.adamw_flat(f.params, f.grads, f.m, f.v, self.lr(self.step_count - 1), self.betas[0], self.betas[1], self.eps,
                       self.wd, self.step_count, grad_scale=1.0 / self.world)
        # the adapter changed: the UNet's INFERENCE plan (bf16 A / B packed at plan time) is stale.  The trainer keeps its own
        # reference (self.P: frozen operands only; its LoRA operands are refreshed by repack()), so this only makes the next
        # pipe(unet=unet) / unet(...) call re-pack -- the in-training validation of [REF train:597-603] then sees the trained LoRA.
        self.unet.invalidate_packed()
        return loss / self.world
