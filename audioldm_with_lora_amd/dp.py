"""Data-parallel runtime for LoRA training: one process per GPU, RCCL over xGMI (torch.distributed backend "nccl").

Mirrors the slice of `accelerate.Accelerator` the reference trainer touches
  [REF script/train/train_audioldm_lora.py:327-332,445-447,494,551,557-561,576,615]:
  init_trackers / prepare / accumulate / backward / clip_grad_norm_ / gather / sync_gradients / log / is_main_process /
  wait_for_everyone / save_state / unwrap_model / end_training / device / num_processes, plus `ProjectConfiguration`.
The reference's DDP traffic collapses to:
  C1  gradient all-reduce  -> ONE all-reduce of the flat fp32 LoRA gradient buffer (<= 7.2 MB at r = 16)
  C2  loss all_gather      -> rides in the extra last slot of the same buffer
  C3  initial broadcast    -> broadcast of the flat LoRA parameter buffer only (base weights load identically per rank)
  C4  barrier              -> dist.barrier
"""
import json
import os
import time
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*); no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return world


def flat_allreduce_mean_(buf: torch.Tensor) -> torch.Tensor:
    """SUM all-reduce of one flat buffer followed by 1/world (DDP gradient semantics).  In place."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        buf.mul_(1.0 / dist.get_world_size())
    return buf


def broadcast_(buf: torch.Tensor, src=0) -> torch.Tensor:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(buf, src=src)
    return buf


def shard_batch(n, rank, world):
    """Contiguous per-rank slice of a global batch of n items (global batch 64 -> 8 per GPU at world 8)."""
    per = n // world
    return slice(rank * per, (rank + 1) * per)


@dataclass
class ProjectConfiguration:
    """accelerate.utils.ProjectConfiguration as the reference builds it [REF train:325]."""
    project_dir: Optional[str] = None
    logging_dir: Optional[str] = None

    def __post_init__(self):
        if self.logging_dir is None:
            self.logging_dir = self.project_dir


class _BatchSamplerShard:
    """accelerate's BatchSamplerShard (split_batches=False, even_batches=True) over the wrapped loader's batch sampler: of every N
    consecutive batches of INDICES process r keeps the r-th -- only the kept batches are ever loaded (wav decoding, mel extraction
    and tokenisation are not repeated N times per rank).  A short last batch is completed, and an incomplete last group of batches is
    filled, by cycling through the epoch's first indices, so that every rank runs the same number of equally sized steps (the flat
    all-reduce averages with a fixed 1 / world).  The shuffle is the SAME permutation on every rank: at the start of each epoch rank
    0's seed is broadcast and seeds the sampler's generator (accelerate: synchronize_rng_states / SeedableRandomSampler)."""

    def __init__(self, batch_sampler, rank, world):
        self.batch_sampler, self.rank, self.world = batch_sampler, rank, world
        self.batch_size = getattr(batch_sampler, "batch_size", None)
        self.drop_last = getattr(batch_sampler, "drop_last", False)

    def __len__(self):
        n = len(self.batch_sampler)
        return n // self.world if self.drop_last else (n + self.world - 1) // self.world

    def _sync_shuffle(self):
        sampler = getattr(self.batch_sampler, "sampler", None)
        if sampler is None or not hasattr(sampler, "generator"):
            return                                           # sequential / custom samplers: nothing random to agree on
        seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            t = seed.cuda() if dist.get_backend() == "nccl" else seed
            dist.broadcast(t, src=0)
            seed = t.cpu()
        g = torch.Generator()
        g.manual_seed(int(seed))
        sampler.generator = g

    def __iter__(self):
        self._sync_shuffle()
        batches = [list(b) for b in self.batch_sampler]      # index lists only: cheap, and every rank derives the same epoch plan
        if not batches:
            return
        head = [i for b in batches for i in b]               # the epoch's indices in order: the pool completions cycle through
        bs = self.batch_size or len(batches[0])
        if len(batches[-1]) < bs and not self.drop_last:
            k = 0
            while len(batches[-1]) < bs:
                batches[-1].append(head[k % len(head)])
                k += 1
        if self.drop_last:
            batches = batches[: len(batches) // self.world * self.world]
        else:
            k = 0
            while len(batches) % self.world:                 # incomplete last group: wrap around to the first batches of the epoch
                batches.append(list(batches[k]))
                k += 1
        for j in range(self.rank, len(batches), self.world):
            yield batches[j]


class ShardedLoader:
    """What `accelerator.prepare(dataloader)` returns under N > 1 processes: accelerate's DataLoaderShard over a BatchSamplerShard with
    its defaults (split_batches=False, even_batches=True).  The loader's batch size stays the PER-PROCESS batch size; of every N
    consecutive batches process r takes the r-th, so the ranks consume disjoint data and the global batch is N x batch_size
    [REF script/train/train_audioldm_lora.py:421-430,445-447].  The sharding happens at the batch SAMPLER (see _BatchSamplerShard):
    a rank loads only its own batches, all ranks shuffle with the same broadcast seed, and short / missing last batches are completed
    from the start of the epoch so that every rank runs the same number of full-size steps."""

    def __init__(self, loader, rank, world):
        self.loader, self.rank, self.world = loader, rank, world
        if getattr(loader, "batch_sampler", None) is None:
            raise ValueError("ShardedLoader needs a DataLoader with a batch sampler (batch_size=None loaders cannot be sharded by batch)")
        self.shard = _BatchSamplerShard(loader.batch_sampler, rank, world)
        kw = dict(num_workers=loader.num_workers, collate_fn=loader.collate_fn, pin_memory=loader.pin_memory, timeout=loader.timeout,
                  worker_init_fn=loader.worker_init_fn, generator=loader.generator, persistent_workers=loader.persistent_workers)
        if loader.num_workers > 0:
            kw.update(prefetch_factor=loader.prefetch_factor, multiprocessing_context=loader.multiprocessing_context)
        self._inner = torch.utils.data.DataLoader(loader.dataset, batch_sampler=self.shard, **kw)

    def __len__(self):
        return len(self.shard)

    def __getattr__(self, name):                             # batch_size, dataset, ... of the wrapped loader
        return getattr(self.loader, name)

    def __iter__(self):
        return iter(self._inner)


class PreparedScheduler:
    """accelerate's AcceleratedScheduler (split_batches=False): one `lr_scheduler.step()` of the loop advances the wrapped schedule
    once PER PROCESS, which is why the reference sizes it with `num_training_steps = max_train_steps * num_processes`
    [REF script/train/train_audioldm_lora.py:438-443,564] -- after k optimiser steps the schedule stands at k N of T N, i.e. at the
    same learning rate as a single process at k of T."""

    def __init__(self, scheduler, num_processes):
        self.scheduler, self.num_processes = scheduler, num_processes

    def step(self, *a, **k):
        for _ in range(self.num_processes):
            self.scheduler.step(*a, **k)

    def __getattr__(self, name):                             # get_last_lr, state_dict, last_epoch, ...
        return getattr(self.scheduler, name)


def _inner_unet(model):
    """PeftModel -> the wrapped UNet (the object that owns the training engine); anything else unchanged."""
    return getattr(getattr(model, "base_model", None), "model", model)


class Accelerator:
    """accelerate-shaped facade over torch.distributed for the HIP trainer.

    Two ways to drive a step:
      * the reference's own loop body [REF train:539-565]: `unet(...)[0]` -> `F.mse_loss` -> `accelerator.backward(loss)` ->
        `optimizer.step()` (optim.AdamW) -- `backward` runs autograd (the HIP launch tape behind training._UNetTrainFn) and then
        ONE all-reduce of the flat LoRA gradient buffer of every prepared model;
      * `training.LoraTrainer.step(...)`: the same arithmetic as one captured hipGraph + one all-reduce + the flat AdamW."""

    def __init__(self, gradient_accumulation_steps=1, mixed_precision=None, log_with=None, project_config=None):
        if gradient_accumulation_steps != 1:
            raise NotImplementedError("the reference trains with gradient_accumulation_steps=1")
        if mixed_precision not in (None, "no"):
            raise NotImplementedError("the reference trains with mixed_precision=None [REF train:329]")
        init_from_env()
        self.distributed = dist.is_available() and dist.is_initialized()
        self.num_processes = dist.get_world_size() if self.distributed else 1
        self.process_index = dist.get_rank() if self.distributed else 0
        self.local_process_index = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = torch.device("cuda", self.local_process_index) if torch.cuda.is_available() else torch.device("cpu")
        self.sync_gradients = True
        self.gradient_accumulation_steps = 1
        self.log_with = log_with                      # "wandb" in the reference; here every tracker is a JSONL file
        self.project_config = project_config or ProjectConfiguration()
        self._models, self._optimizers, self._schedulers = [], [], []
        self._log_file = None
        self._log_path = None

    @property
    def is_main_process(self):
        return self.process_index == 0

    @property
    def is_local_main_process(self):
        return self.local_process_index == 0

    # ---- tracking (wandb in the reference [REF train:333-346,570,583-589]; JSONL here: no network) ----
    def init_trackers(self, project_name, config=None, init_kwargs=None):
        if not self.is_main_process:
            return
        d = self.project_config.logging_dir or "."
        os.makedirs(d, exist_ok=True)
        self._log_path = os.path.join(d, f"{project_name}.metrics.jsonl")
        self._log_file = open(self._log_path, "a")
        run = (init_kwargs or {}).get("wandb", {})
        self._log_file.write(json.dumps({"event": "init", "project": project_name, "time": time.time(),
                                         "run": {k: run[k] for k in run if isinstance(run[k], (str, int, float, list))}}) + "\n")
        self._log_file.flush()

    def log(self, values, step=None):
        if not self.is_main_process or self._log_file is None:
            return
        rec = {"step": step}
        for k, v in values.items():
            rec[k] = float(v) if isinstance(v, (int, float)) or (torch.is_tensor(v) and v.numel() == 1) else str(v)
        self._log_file.write(json.dumps(rec) + "\n")
        self._log_file.flush()

    # ---- model / optimiser plumbing ----
    def prepare(self, *objs):
        """Registers what the loop will use and returns the objects the loop must use from here on, in the same order
        [REF train:445-447].  A LoRA-wrapped UNet gets its training engine here: the LoRA parameters move into one flat fp32
        buffer and rank 0's copy is broadcast (DDP's constructor broadcast, C3, LoRA buffer only).  Under N > 1 processes a
        DataLoader comes back rank-sharded (ShardedLoader) and the LR scheduler comes back stepping N times per call
        (PreparedScheduler), exactly as accelerate prepares them; with one process both come back unchanged."""
        from . import optim
        out = []
        for o in objs:
            inner = _inner_unet(o)
            if isinstance(o, torch.nn.Module) and hasattr(inner, "_has_trainable_lora") and inner._has_trainable_lora():
                if inner.conv_in.weight.is_cuda:
                    from .training import trainer_of
                    trainer_of(inner)
                self._models.append(o)
            elif isinstance(o, optim.AdamW):
                self._optimizers.append(o)
            elif isinstance(o, optim.PolynomialLR):
                self._schedulers.append(o)
                if self.num_processes > 1:
                    o = PreparedScheduler(o, self.num_processes)
            elif isinstance(o, torch.utils.data.DataLoader) and self.num_processes > 1:
                o = ShardedLoader(o, self.process_index, self.num_processes)
            out.append(o)
        return tuple(out) if len(out) != 1 else out[0]

    def unwrap_model(self, model):
        """accelerate strips only the distributed wrapper (`.module`); a PeftModel stays a PeftModel [REF train:347-350,577,598]."""
        return getattr(model, "module", model)

    @contextmanager
    def accumulate(self, model):
        yield

    def backward(self, loss, **kw):
        """[REF train:557]: autograd backward, then DDP's gradient all-reduce -- one collective per prepared model."""
        loss.backward(**kw)
        from .training import trainer_of
        for m in self._models:
            tr = trainer_of(_inner_unet(m), create=False)
            if tr is not None:
                tr.allreduce_grads_()

    def clip_grad_norm_(self, parameters, max_norm, norm_type=2):
        """[REF train:559-561].  In the reference `parameters` is an already-exhausted iterator, so the call clips nothing
        (SURVEY.md quirk Q1); with a real parameter list this clips the gradients in place like torch's utility."""
        params = [p for p in parameters if getattr(p, "grad", None) is not None]
        if not params:
            return torch.zeros((), device=self.device)
        if norm_type != 2:
            raise NotImplementedError("only the 2-norm is implemented")
        total = torch.sqrt(sum((p.grad.detach().float() ** 2).sum() for p in params))
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        for p in params:
            p.grad.mul_(coef)
        return total

    def gather(self, t):
        if not self.distributed or self.num_processes == 1:
            return t.reshape(1) if t.dim() == 0 else t
        t = t.detach()
        out = [torch.empty_like(t) for _ in range(self.num_processes)]
        dist.all_gather(out, t)
        return torch.stack(out) if t.dim() == 0 else torch.cat(out)

    def wait_for_everyone(self):
        if self.distributed and self.num_processes > 1:
            dist.barrier()

    def save_state(self, output_dir, trainer=None):
        """[REF train:574-576] `accelerator.save_state(save_path)`: LoRA-only `model.safetensors` with peft key names (what the
        reference evidently intended) + `optimizer.bin` + `scheduler.bin`.  Takes the models /
        optimisers registered by prepare(); `trainer=` adds a LoraTrainer (the fast path keeps its moments there)."""
        if not self.is_main_process:
            return
        models = [_inner_unet(m) for m in self._models]
        if trainer is not None and all(trainer.unet is not m for m in models):
            models.append(trainer.unet)
        if not models:
            raise RuntimeError("Accelerator.save_state: nothing to save -- pass the model through accelerator.prepare(...) first "
                               "(or give save_state the LoraTrainer)")
        from safetensors.torch import save_file
        os.makedirs(output_dir, exist_ok=True)
        for i, m in enumerate(models):
            sd = {("base_model.model." + n): p.detach().float().cpu().contiguous() for n, p in m.named_parameters() if "lora_" in n}
            save_file(sd, os.path.join(output_dir, "model.safetensors" if i == 0 else f"model_{i}.safetensors"))
        if trainer is not None:
            torch.save({"m": trainer.flat.m.cpu(), "v": trainer.flat.v.cpu(), "step": trainer.step_count},
                       os.path.join(output_dir, "optimizer.bin"))
        for i, o in enumerate(self._optimizers):
            torch.save(o.state_dict(), os.path.join(output_dir, "optimizer.bin" if (i == 0 and trainer is None) else f"optimizer_{i}.bin"))
        for i, sch in enumerate(self._schedulers):
            torch.save(sch.state_dict(), os.path.join(output_dir, "scheduler.bin" if i == 0 else f"scheduler_{i}.bin"))

    def end_training(self):
        self.wait_for_everyone()
        if self._log_file is not None:
            self._log_file.close()
            self._log_file = None
