"""GPU: the PRODUCT's data-parallel code with two ranks.

Two fresh child processes share cuda:0 and rendezvous over gloo (CUDA tensors; one gpurun box has one GPU, so RCCL itself cannot
be exercised here -- what runs is everything around the collective call: the DDP-style LoRA broadcast at engine construction,
`LoraTrainer.step` with world > 1 (ONE all-reduce of the flat gradient buffer + loss slot, grad_scale = 1 / world),
`LoraTrainer.allreduce_grads_` behind `Accelerator.backward`, the rank-sharded DataLoader and the N-times-stepping LR schedule of
`Accelerator.prepare`).  Each rank trains on its half of the global batch; the result must equal a single process on the whole
batch: [REF script/train/train_audioldm_lora.py:445-447,557,563-565]."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

STEPS, GLOBAL_B, LR0, MAX_STEPS = 2, 4, 1.0e-3, 20


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _unet(seed=0):
    from audioldm_with_lora_amd import lora as plora
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    from oracle import configs
    torch.manual_seed(seed)
    unet = UNet2DConditionModel(**configs.tiny_unet())
    unet.requires_grad_(False)
    punet = plora.get_peft_model(unet, plora.LoraConfig(r=2, lora_alpha=2, target_modules=["to_q", "to_k", "to_v", "to_out.0"],
                                                        init_lora_weights="gaussian"))
    g = torch.Generator().manual_seed(seed + 1)
    sd = punet.state_dict()
    for k in sd:
        if "lora_B" in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
    punet.load_state_dict(sd)
    return punet, unet


def _data():
    g = torch.Generator().manual_seed(11)
    n = STEPS * GLOBAL_B
    return dict(latents=torch.randn(n, 8, 16, 16, generator=g) * 0.92, noise=torch.randn(n, 8, 16, 16, generator=g),
                timesteps=torch.randint(0, 1000, (n,), generator=g), prompt_embeds=F.normalize(torch.randn(n, 64, generator=g), dim=-1))


class _Items(torch.utils.data.Dataset):
    def __init__(self, d):
        self.d = d

    def __len__(self):
        return self.d["latents"].shape[0]

    def __getitem__(self, i):
        return {k: v[i] for k, v in self.d.items()}


def _run_trainer(rank, world):
    """LoraTrainer.step on this rank's contiguous slice of every global batch"""
    from audioldm_with_lora_amd import dp
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    from audioldm_with_lora_amd.training import LoraTrainer
    punet, unet = _unet()
    unet.to("cuda")
    if rank == 1:                                              # DDP's constructor broadcast must overwrite this
        with torch.no_grad():
            for n, p in unet.named_parameters():
                if "lora_A" in n:
                    p.add_(1.0)
    tr = LoraTrainer(unet, DDIMScheduler(), lr=LR0, max_train_steps=MAX_STEPS, use_graph=False)
    start = tr.flat.params.detach().clone()
    d, losses = _data(), []
    for s in range(STEPS):
        sl = dp.shard_batch(GLOBAL_B, rank, world)
        lo = s * GLOBAL_B
        idx = slice(lo + sl.start, lo + sl.stop)
        losses.append(float(tr.step(d["latents"][idx], d["noise"][idx], d["timesteps"][idx], d["prompt_embeds"][idx])))
    return dict(start=start.cpu(), params=tr.flat.params.detach().cpu(), losses=losses, lr=tr.lr(tr.step_count))


def _run_accelerate(rank, world):
    """the reference loop's calls: prepare -> unet(...)[0] -> mse -> accelerator.backward -> optimizer.step -> lr_scheduler.step"""
    from audioldm_with_lora_amd import dp, optim
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    from audioldm_with_lora_amd.training import trainer_of
    acc = dp.Accelerator()
    assert acc.num_processes == world
    punet, unet = _unet()
    punet.to(acc.device)
    opt = optim.AdamW([p for p in punet.parameters() if p.requires_grad], lr=LR0, betas=(0.9, 0.999), weight_decay=1e-5, eps=1e-8)
    sch = optim.get_scheduler("polynomial", optimizer=opt, num_warmup_steps=0, num_training_steps=MAX_STEPS * acc.num_processes)
    loader = torch.utils.data.DataLoader(_Items(_data()), batch_size=GLOBAL_B // world, shuffle=False)
    punet, opt, loader, sch = acc.prepare(punet, opt, loader, sch)
    assert len(loader) == STEPS
    ddim, losses, first_ids = DDIMScheduler(), [], []
    punet.train()
    opt.zero_grad()
    for batch in loader:
        lat, noise = batch["latents"].to(acc.device), batch["noise"].to(acc.device)
        t = batch["timesteps"].to(acc.device).long()
        first_ids.append(float(batch["latents"][0, 0, 0, 0]))
        pred = punet(ddim.add_noise(lat, noise, t), t, encoder_hidden_states=None, class_labels=batch["prompt_embeds"].to(acc.device),
                     return_dict=False)[0]
        loss = F.mse_loss(pred.float(), noise.float(), reduction="mean")
        losses.append(float(acc.gather(loss).mean()))
        acc.backward(loss)
        opt.step()
        sch.step()
        opt.zero_grad()
    acc.wait_for_everyone()
    flat = trainer_of(unet, create=False).flat
    return dict(params=flat.params.detach().cpu(), losses=losses, lr=sch.get_last_lr()[0], first=first_ids)


def _worker(rank, world, port, out, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    import torch.distributed as dist
    from audioldm_with_lora_amd import dp
    assert dp.init_from_env(backend="gloo") == world            # gloo over CUDA tensors: both ranks live on the one GPU of this box
    res = _run_trainer(rank, world) if mode == "trainer" else _run_accelerate(rank, world)
    torch.save(res, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def _rel(a, b):
    return float((a - b).norm() / b.norm())


def _spawn(tmp_path, mode):
    out = str(tmp_path / f"{mode}.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, mode), nprocs=2, join=True)
    return torch.load(out + ".0"), torch.load(out + ".1")


def test_lora_trainer_step_two_ranks_equals_one_rank_on_the_whole_batch(tmp_path):
    r0, r1 = _spawn(tmp_path, "trainer")
    one = _run_trainer(0, 1)                                    # this process: no process group, the whole batch
    # C3: rank 1 started from rank 0's adapter, not from its own perturbed one
    assert torch.equal(r0["start"], r1["start"]) and torch.equal(r0["start"], one["start"])
    # the ranks stay in lock step: identical parameters after identical (all-reduced) gradients
    assert torch.equal(r0["params"], r1["params"])
    # grad_scale = 1 / world and ONE sum: the update of the mean gradient, i.e. the single-process update
    assert _rel(r0["params"], one["params"]) < 1e-3
    d2, d1 = r0["params"] - r0["start"], one["params"] - one["start"]
    assert float(d2.norm()) > 0 and float((d2 * d1).sum() / (d2.norm() * d1.norm())) > 0.99
    # the loss slot rides in the same collective and comes back as the all-rank mean
    assert r0["losses"] == r1["losses"]
    for a, b in zip(r0["losses"], one["losses"]):
        assert abs(a - b) < 2e-3 * abs(b) + 1e-6, (r0["losses"], one["losses"])
    assert r0["lr"] == one["lr"]


def test_accelerator_prepare_backward_two_ranks_equals_one_rank(tmp_path):
    r0, r1 = _spawn(tmp_path, "accel")
    one = _run_accelerate(0, 1)
    assert r0["first"] != r1["first"], "the prepared DataLoader must hand the two ranks different batches"
    assert r0["first"] == one["first"]                          # rank 0's batches open the global batches
    assert torch.equal(r0["params"], r1["params"])
    assert _rel(r0["params"], one["params"]) < 1e-3
    for a, b in zip(r0["losses"], one["losses"]):              # gather(loss).mean() == the whole batch's loss
        assert abs(a - b) < 2e-3 * abs(b) + 1e-6, (r0["losses"], one["losses"])
    assert abs(r0["lr"] - one["lr"]) < 1e-15 and r0["lr"] == r1["lr"]
