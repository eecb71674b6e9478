"""The DDIM denoising loop as a replayed hipGraph.

AudioLDMPipeline.__call__ step 5 ([REF script/inference/generate_audio.py:47-52], [REF app.py:14]):
    for t in timesteps:  eps = unet(cat[x, x], t, class_labels=[neg | pos]);  eps = eps_u + g (eps_t - eps_u)
                         x = ddim_step(eps, t, x)
One step is ~450 kernel launches of a few microseconds each, so the loop is launch-bound when driven from
the host.  Everything that changes between steps lives in DEVICE memory -- the latent x, the CFG-doubled
bf16 UNet input, the timestep scalar and a step counter that indexes the precomputed DDIM coefficient
table -- so ONE captured graph of a single step replays unchanged for all N steps with no host work
in between.
"""
import torch

from . import ops


class DenoiseEngine:
    def __init__(self, unet, scheduler, batch, height, width, num_inference_steps, guidance_scale=2.5,
                 device="cuda", use_graph=True, chains=None):
        self.unet, self.scheduler = unet, scheduler
        self.B, self.H, self.W = batch, height, width
        self.C = unet.cfg["in_channels"]
        self.cfg = guidance_scale > 1.0
        self.g = float(guidance_scale)
        self.n_steps = num_inference_steps
        self.use_graph = use_graph
        dev = torch.device(device)
        self.dev = dev
        scheduler.set_timesteps(num_inference_steps)
        self.timesteps_f32 = scheduler.timesteps.to(torch.float32).to(dev)
        self.coef = scheduler.coefficient_table().contiguous().to(dev)
        # Optional: independent sub-batches ("chains") captured as parallel branches of the graph (each owns a contiguous
        # slice of the latents and its own CFG-doubled input block [uncond_i | cond_i]).  Measured on MI355X / ROCm 7.2 at
        # batch 4: 1 chain 4.83 ms/step, 2 chains 4.81, 4 chains 5.90 -- the branches do not overlap usefully, so the
        # default stays a single chain.
        if chains is None:
            chains = 1
        assert batch % chains == 0
        self.chains, self.bc = chains, batch // chains
        nbc = 2 * self.bc if self.cfg else self.bc
        self.x = torch.zeros(batch, height, width, self.C, dtype=torch.float32, device=dev)       # latents, NHWC fp32
        self.x_in = [torch.zeros(nbc, height, width, self.C, dtype=torch.bfloat16, device=dev) for _ in range(chains)]
        self.t_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_idx = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)      # aldm_ddim_step_fused's last-workgroup ticket (rests at 0)
        self.cls = None
        self.temb = None             # [chains][n_steps, nbc, temb_total] fp32: time-embedding projections of every step
        self.rowbias = None          # [chains][nbc, temb_total] fp32: the current step's row (gathered on the device)
        self.graph = None
        self._side = None
        # The captured graph holds raw pointers into the UNet's packed operands (unet.plan()): keep that plan alive for as long
        # as the graph exists and remember its version, so a later state load / LoRA update (which bumps unet.plan_version) is
        # noticed instead of replaying the old weights -- or freed memory.
        self._plan_ref = None
        self.plan_version = None

    def stale(self):
        """True when the UNet's packed operands changed after this engine captured its graph."""
        return self.plan_version is not None and self.plan_version != self.unet.plan_version

    def _check_fresh(self):
        if self.stale():
            if self.graph is None:                     # eager launches re-plan by themselves
                self._plan_ref, self.plan_version = self.unet.plan(), self.unet.plan_version
                return
            raise ops._lib.AldmError("DenoiseEngine: the UNet's weights changed after this graph was captured "
                                     "(load_state_dict / LoRA update); call capture() again or build a new engine")

    def set_condition(self, prompt_embeds, negative_prompt_embeds=None):
        """[B, D] L2-normalised prompt embeddings (CLAP text_embeds); CFG order is [negative | positive]."""
        pe = prompt_embeds.to(self.dev, torch.float32)
        ne = None
        if self.cfg:
            ne = torch.zeros_like(pe) if negative_prompt_embeds is None else negative_prompt_embeds.to(self.dev, torch.float32)
        new = []
        for i in range(self.chains):
            sl = slice(i * self.bc, (i + 1) * self.bc)
            e = torch.cat([ne[sl], pe[sl]]) if self.cfg else pe[sl]
            new.append(ops.f32_to_bf16(e.contiguous()))
        if self.cls is None:
            self.cls = new
        else:
            for dst, src in zip(self.cls, new):
                dst.copy_(src)
        # everything the UNet derives from (timestep, prompt) alone is computed here once for the whole schedule
        tabs = [self.unet.temb_table(self.timesteps_f32, c) for c in self.cls]
        if self.temb is None:
            self.temb = tabs
            self.rowbias = [torch.empty_like(t[0]) for t in tabs]
        else:
            for dst, src in zip(self.temb, tabs):
                dst.copy_(src)
        self._prime()

    def set_latents(self, latents_nchw):
        """latents [B, C, H, W] fp32 (already multiplied by init_noise_sigma = 1)."""
        x = ops.nchw_to_nhwc(latents_nchw.to(self.dev, torch.float32).contiguous(), out_f32=True)
        self.x.copy_(x)
        xb = ops.f32_to_bf16(self.x)
        for i in range(self.chains):
            sl = slice(i * self.bc, (i + 1) * self.bc)
            self.x_in[i][: self.bc].copy_(xb[sl])
            if self.cfg:
                self.x_in[i][self.bc:].copy_(xb[sl])
        self.step_idx.zero_()
        self.t_buf.copy_(self.timesteps_f32[:1])
        self._prime()

    def _prime(self):
        """The single-chain step gathers the NEXT step's time-embedding row at its end (ops.ddim_step_fused); the row of the
        step the counter stands at is put in place here, whenever the counter or the table changes outside the graph.  The fused
        step's last-workgroup ticket is put back to rest as well: a launch that was aborted midway would otherwise leave it non-zero
        for every later replay."""
        self.ticket.zero_()
        if self.temb is not None and self.chains == 1:
            ops.gather_row(self.temb[0], self.step_idx, self.rowbias[0])

    def _chain_step(self, i):
        ops.gather_row(self.temb[i], self.step_idx, self.rowbias[i])
        eps = self.unet.forward_nhwc(self.x_in[i], self.t_buf, self.cls[i], rowbias=self.rowbias[i])
        ops.cfg_ddim_step(eps, self.x[i * self.bc:(i + 1) * self.bc], self.cfg, self.g, self.coef, self.step_idx, self.x_in[i])

    def _one_step(self):
        if self.chains == 1:
            # one chain: guidance + DDIM update, the next step's time-embedding row and the step counter in ONE launch behind the UNet
            eps = self.unet.forward_nhwc(self.x_in[0], self.t_buf, self.cls[0], rowbias=self.rowbias[0])
            ops.ddim_step_fused(eps, self.x, self.cfg, self.g, self.coef, self.step_idx, self.x_in[0], self.temb[0], self.rowbias[0],
                                self.timesteps_f32, self.t_buf, self.ticket)
            return
        else:                                   # fork / join: under capture these become parallel graph branches
            cur = torch.cuda.current_stream()
            if self._side is None:
                self._side = [torch.cuda.Stream() for _ in range(self.chains - 1)]
            for s in self._side:
                s.wait_stream(cur)
            self._chain_step(0)
            for i, s in enumerate(self._side):
                with torch.cuda.stream(s):
                    self._chain_step(i + 1)
            for s in self._side:
                cur.wait_stream(s)
        ops.advance_step(self.step_idx, self.timesteps_f32, self.t_buf)

    def capture(self):
        """Warm up (loads code objects, sizes the split-K workspace) and capture one step."""
        self.graph = None
        self._plan_ref, self.plan_version = self.unet.plan(), self.unet.plan_version
        saved = (self.x.clone(), [t.clone() for t in self.x_in], self.step_idx.clone(), self.t_buf.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._one_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if self.use_graph:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):   # RCCL's watchdog thread must not void the capture
                self._one_step()
        for dst, src in zip([self.x] + self.x_in + [self.step_idx, self.t_buf], [saved[0]] + saved[1] + [saved[2], saved[3]]):
            dst.copy_(src)
        self._prime()
        torch.cuda.synchronize()

    def step(self):
        self._check_fresh()
        if self.graph is not None:
            self.graph.replay()
        else:
            self._one_step()

    def run(self, steps=None):
        for _ in range(self.n_steps if steps is None else steps):
            self.step()
        return self.x

    def latents_nchw(self):
        return ops.nhwc_to_nchw_f32(self.x)
