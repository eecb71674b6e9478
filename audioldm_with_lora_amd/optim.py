"""Optimiser and LR schedule of the reference trainer, on the HIP path.

Mirrors what the reference constructs at  [REF script/train/train_audioldm_lora.py:396-403]  (`torch.optim.AdamW(lora_layers,
lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-5, eps=1e-8)`) and  [REF train:438-443]  (`diffusers.optimization.get_scheduler(
"polynomial", optimizer=, num_warmup_steps=0, num_training_steps=)`), and drives at  [REF train:563-565]  (`optimizer.step();
lr_scheduler.step(); optimizer.zero_grad()`).

The parameters are the LoRA A / B matrices.  Once the training engine exists (training.LoraTrainer -- created by
`Accelerator.prepare` or by the first training-mode `unet(...)` call) they are views into ONE flat fp32 buffer, their `.grad`s
views into one flat gradient buffer, and `step()` is a single `aldm_adamw_flat` launch over the whole buffer (decoupled weight
decay, bias-corrected moments: the arithmetic of torch.optim.AdamW, tested against it in tests/test_gpu_training.py).
"""
import torch

from . import ops
from ._lib import AldmError


class AdamW:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.param_groups = [dict(self.defaults, params=self.params, initial_lr=lr)]
        self.state = {}                      # per-parameter moments for parameters that are NOT in a flat buffer
        self._step = 0

    # ---- torch.optim.Optimizer surface the reference loop touches ----
    def zero_grad(self, set_to_none=True):
        """The flat gradient buffer is cleared by the next forward (the engine zeroes it before the tape runs), so dropping
        the views is all there is to do; set_to_none=False zeroes them in place."""
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _flat_of(self):
        """(FlatLora, covers_everything) when every parameter is a view into the same flat buffer, else (None, False)."""
        flats = {id(getattr(p, "_aldm_flat", (None,))[0]) for p in self.params}
        f = getattr(self.params[0], "_aldm_flat", (None,))[0]
        if f is None or len(flats) != 1:
            return None, False
        return f, sum(p.numel() for p in self.params) == f.n

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not used by the reference trainer")
        g = self.param_groups[0]
        self._step += 1
        b1, b2 = g["betas"]
        flat, whole = self._flat_of()
        if flat is not None and whole:
            ops.adamw_flat(flat.params, flat.grads, flat.m, flat.v, g["lr"], b1, b2, g["eps"], g["weight_decay"], self._step,
                           grad_scale=1.0)
            flat.owner_changed()
            return
        for p in self.params:                # parameters outside a flat buffer: the same kernel, one launch per tensor
            if p.grad is None:
                continue
            if not p.is_cuda:
                raise AldmError("AdamW: parameters must live on the MI355X (no CPU fallback)")
            st = self.state.setdefault(id(p), dict(m=torch.zeros_like(p, dtype=torch.float32), v=torch.zeros_like(p, dtype=torch.float32)))
            pd, gd = p.data, p.grad
            if pd.dtype != torch.float32 or not pd.is_contiguous() or not gd.is_contiguous():
                raise AldmError("AdamW: fp32 contiguous parameters only (LoRA matrices are fp32 trainables, SURVEY.md B.7)")
            ops.adamw_flat(pd.view(-1), gd.reshape(-1).float(), st["m"].view(-1), st["v"].view(-1), g["lr"], b1, b2, g["eps"],
                           g["weight_decay"], self._step, grad_scale=1.0)
        if flat is not None:
            flat.owner_changed()

    def state_dict(self):
        flat, whole = self._flat_of()
        sd = {"step": self._step, "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}
        if flat is not None and whole:
            sd["m"], sd["v"] = flat.m.detach().cpu(), flat.v.detach().cpu()
        if self.state:                                       # parameters outside a flat buffer keep their moments here
            index = {id(p): i for i, p in enumerate(self.params)}
            sd["state"] = {index[k]: {"m": st["m"].detach().cpu(), "v": st["v"].detach().cpu()} for k, st in self.state.items()}
        return sd


class PolynomialLR:
    """diffusers.optimization.get_polynomial_decay_schedule_with_warmup(lr_end=1e-7, power=1.0): the lambda of
    transformers/optimization.py (pinned in tests/golden/poly_lr.npz)."""

    def __init__(self, optimizer, num_warmup_steps, num_training_steps, lr_end=1e-7, power=1.0):
        self.optimizer, self.warm, self.total, self.lr_end, self.power = optimizer, num_warmup_steps, num_training_steps, lr_end, power
        self.lr_init = optimizer.param_groups[0]["initial_lr"]
        if not self.lr_init > lr_end:
            raise ValueError(f"lr_end ({lr_end}) must be smaller than initial lr ({self.lr_init})")
        self.last_epoch = 0
        self._apply()

    def lr_at(self, step):
        if step < self.warm:
            return self.lr_init * step / max(1, self.warm)
        if step > self.total:
            return self.lr_end
        remaining = 1 - (step - self.warm) / (self.total - self.warm)
        return (self.lr_init - self.lr_end) * remaining ** self.power + self.lr_end

    def _apply(self):
        for g in self.optimizer.param_groups:
            g["lr"] = self.lr_at(self.last_epoch)

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_epoch": self.last_epoch}


def get_scheduler(name, optimizer, num_warmup_steps=0, num_training_steps=None, lr_end=1e-7, power=1.0, **kw):
    if name != "polynomial":
        raise NotImplementedError(f"lr schedule {name!r}: the reference trains with 'polynomial' [REF train:438-443]")
    return PolynomialLR(optimizer, num_warmup_steps, num_training_steps, lr_end, power)
