"""GPU parity: VAE decode / encode, HiFi-GAN vocoder and the whole AudioLDMPipeline vs the CPU oracle
(and the vocoder against the transformers-derived golden vectors)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    import conftest
    return conftest.record(float((a - b).norm() / b.norm()))


def test_vocoder_matches_transformers_golden():
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    from oracle import configs
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "vocoder_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    m = SpeechT5HifiGan(**configs.tiny_vocoder())
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    got = m(torch.from_numpy(z["mel"]).cuda()).cpu()
    want = torch.from_numpy(z["wav"])
    assert got.shape == want.shape == (2, 160 * 12 + 32)
    assert rel_l2(got, want) < 3e-2 and float((got - want).abs().max()) < 3e-2


def test_vocoder_full_config_matches_oracle():
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    from oracle.hifigan import SpeechT5HifiGan as OVoc
    torch.manual_seed(11)
    ref = OVoc().eval()
    g = torch.Generator().manual_seed(12)
    sd = ref.state_dict()
    for k, v in sd.items():          # O(1) activations through the stack
        if k.endswith("weight"):
            fan_in = v[0].numel() if "upsampler" not in k else v.shape[0] * v.shape[2] / 2
            v.copy_(torch.randn(v.shape, generator=g) * (1.0 / fan_in) ** 0.5)
    ref.load_state_dict(sd)
    mine = SpeechT5HifiGan()
    mine.load_state_dict(sd, strict=True)
    mine = mine.cuda()
    mel = torch.randn(1, 24, 64, generator=g)
    with torch.no_grad():
        want = ref(mel)
    got = mine(mel.cuda()).cpu()
    assert got.shape == want.shape == (1, 160 * 24 + 32)
    assert rel_l2(got, want) < 4e-2, rel_l2(got, want)


@pytest.mark.parametrize("cfgname,hw", [("tiny", (6, 4)), ("tiny", (25, 16)), ("full", (12, 16))])
def test_vae_decode_matches_oracle(cfgname, hw):
    from audioldm_with_lora_amd.vae import AutoencoderKL
    from oracle import configs
    from oracle.vae import AutoencoderKL as OVae
    cfg = configs.tiny_vae() if cfgname == "tiny" else {}
    torch.manual_seed(21)
    ref = OVae(**cfg).eval()
    mine = AutoencoderKL(**cfg)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.cuda()
    z = torch.randn(2, 8, *hw, generator=torch.Generator().manual_seed(22))
    with torch.no_grad():
        want = ref.decode(z).sample
    got = mine.decode(z.cuda()).sample.cpu()
    assert got.shape == want.shape == (2, 1, 4 * hw[0], 4 * hw[1])
    assert rel_l2(got, want) < 4e-2, rel_l2(got, want)


def test_vae_encode_matches_oracle():
    from audioldm_with_lora_amd.vae import AutoencoderKL
    from oracle import configs
    from oracle.vae import AutoencoderKL as OVae
    torch.manual_seed(23)
    ref = OVae(**configs.tiny_vae()).eval()
    mine = AutoencoderKL(**configs.tiny_vae())
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.cuda()
    x = torch.randn(2, 1, 40, 16, generator=torch.Generator().manual_seed(24))
    with torch.no_grad():
        want = ref.encode(x).latent_dist
    got = mine.encode(x.cuda()).latent_dist
    assert got.mean.shape == want.mean.shape == (2, 8, 10, 4)
    assert rel_l2(got.mean.cpu(), want.mean) < 4e-2
    assert rel_l2(got.std.cpu(), want.std) < 4e-2
    # latent_dist.sample() [REF train:495] is one kernel: mean + exp(0.5 clamp(logvar)) * noise with the generator's noise
    s1 = got.sample(torch.Generator(device="cuda").manual_seed(3))
    noise = torch.randn(got.mean.shape, generator=torch.Generator(device="cuda").manual_seed(3), dtype=torch.float32, device="cuda")
    torch.testing.assert_close(s1.float(), got.mean.float() + got.std.float() * noise, rtol=1e-5, atol=1e-6)


def test_pipeline_end_to_end_config1_shape_tiny_models():
    """Plumbing check of steps 1-8 with shrunken models: prompt_embeds -> DDIM loop -> VAE -> vocoder -> trim."""
    from audioldm_with_lora_amd.pipeline import AudioLDMPipeline
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    from audioldm_with_lora_amd.vae import AutoencoderKL
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    from oracle import configs
    from oracle.ddim import DDIMScheduler as ODDIM
    from oracle.hifigan import SpeechT5HifiGan as OVoc
    from oracle.pipeline import AudioLDMPipeline as OPipe
    from oracle.unet import UNet2DConditionModel as OUNet
    from oracle.vae import AutoencoderKL as OVae
    torch.manual_seed(31)
    ou, ov, oh = OUNet(**configs.tiny_unet()).eval(), OVae(**configs.tiny_vae()).eval(), OVoc(**configs.tiny_vocoder()).eval()
    g = torch.Generator().manual_seed(32)
    sd = oh.state_dict()
    for k, v in sd.items():
        if k.endswith("weight"):
            fan_in = v[0].numel() if "upsampler" not in k else v.shape[0] * v.shape[2] / 2
            v.copy_(torch.randn(v.shape, generator=g) * (1.0 / fan_in) ** 0.5)
    oh.load_state_dict(sd)
    opipe = OPipe(ou, ov, oh, ODDIM())
    u, v, h = UNet2DConditionModel(**configs.tiny_unet()), AutoencoderKL(**configs.tiny_vae()), SpeechT5HifiGan(**configs.tiny_vocoder())
    u.load_state_dict(ou.state_dict()); v.load_state_dict(ov.state_dict()); h.load_state_dict(oh.state_dict())
    pipe = AudioLDMPipeline(v, None, None, u, DDIMScheduler(), h).to("cuda")
    pe = torch.nn.functional.normalize(torch.randn(1, 64, generator=g), dim=-1)
    ne = torch.nn.functional.normalize(torch.randn(1, 64, generator=g), dim=-1)
    lat = torch.randn(1, 8, 32, 16, generator=g)                          # 1.28 s clip
    want = opipe(pe, ne, audio_length_in_s=1.28, num_inference_steps=6, guidance_scale=2.5, latents=lat.clone())
    got = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, audio_length_in_s=1.28, num_inference_steps=6,
               guidance_scale=2.5, latents=lat.clone())
    assert got.audios.shape == want.audios.shape == (1, 20480)
    a, b = torch.from_numpy(got.audios), torch.from_numpy(want.audios)
    assert torch.isfinite(a).all()
    assert rel_l2(a, b) < 8e-2, rel_l2(a, b)


def test_pipeline_from_prompt_through_the_clap_tower():
    """prompt (strings) -> tokenizer (host stub with the RoBERTa contract) -> CLAP text tower on the HIP kernels ->
    normalised embeddings -> loop -> decode, against the oracle pipeline fed the oracle CLAP tower's embeddings
    [REF script/inference/generate_audio.py:42-52]."""
    from types import SimpleNamespace
    from audioldm_with_lora_amd.clap_text import ClapTextModelWithProjection
    from audioldm_with_lora_amd.pipeline import AudioLDMPipeline
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    from audioldm_with_lora_amd.vae import AutoencoderKL
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    from oracle import configs
    from oracle.clap_text import ClapTextModelWithProjection as OClap
    from oracle.ddim import DDIMScheduler as ODDIM
    from oracle.hifigan import SpeechT5HifiGan as OVoc
    from oracle.pipeline import AudioLDMPipeline as OPipe
    from oracle.unet import UNet2DConditionModel as OUNet
    from oracle.vae import AutoencoderKL as OVae
    torch.manual_seed(41)
    ccfg = dict(configs.tiny_clap_text(), projection_dim=64, max_position_embeddings=80)     # UNet class-embedding input = 64
    ou, ov, oh = OUNet(**configs.tiny_unet()).eval(), OVae(**configs.tiny_vae()).eval(), OVoc(**configs.tiny_vocoder()).eval()
    oc = OClap(**ccfg).eval()
    g = torch.Generator().manual_seed(42)
    sd = oh.state_dict()
    for k, v in sd.items():
        if k.endswith("weight"):
            fan_in = v[0].numel() if "upsampler" not in k else v.shape[0] * v.shape[2] / 2
            v.copy_(torch.randn(v.shape, generator=g) * (1.0 / fan_in) ** 0.5)
    oh.load_state_dict(sd)
    u, v, h = UNet2DConditionModel(**configs.tiny_unet()), AutoencoderKL(**configs.tiny_vae()), SpeechT5HifiGan(**configs.tiny_vocoder())
    c = ClapTextModelWithProjection(**ccfg)
    u.load_state_dict(ou.state_dict()); v.load_state_dict(ov.state_dict()); h.load_state_dict(oh.state_dict()); c.load_state_dict(oc.state_dict())

    class StubTokenizer:                 # same call contract as RobertaTokenizerFast: <s> ids </s> <pad>...
        model_max_length = 64

        def __call__(self, text, padding=None, max_length=None, truncation=None, return_tensors=None):
            ids = torch.full((len(text), max_length), 1, dtype=torch.long)
            mask = torch.zeros(len(text), max_length, dtype=torch.long)
            for i, t in enumerate(text):
                toks = [0] + [3 + (ord(ch) % 190) for ch in t][: max_length - 2] + [2]
                ids[i, : len(toks)] = torch.tensor(toks)
                mask[i, : len(toks)] = 1
            return SimpleNamespace(input_ids=ids, attention_mask=mask)

    tok = StubTokenizer()
    prompts = ["a dog barking in the rain", "boom bap drums"]
    t_pos, t_neg = tok(prompts, max_length=64), tok(["", ""], max_length=64)
    pe = torch.nn.functional.normalize(oc(t_pos.input_ids, t_pos.attention_mask).text_embeds, dim=-1)
    ne = torch.nn.functional.normalize(oc(t_neg.input_ids, t_neg.attention_mask).text_embeds, dim=-1)
    lat = torch.randn(2, 8, 32, 16, generator=g)
    want = OPipe(ou, ov, oh, ODDIM())(pe, ne, audio_length_in_s=1.28, num_inference_steps=4, guidance_scale=2.5, latents=lat.clone())
    pipe = AudioLDMPipeline(v, c, tok, u, DDIMScheduler(), h).to("cuda")
    got = pipe(prompt=prompts, audio_length_in_s=1.28, num_inference_steps=4, guidance_scale=2.5, latents=lat.clone())
    assert got.audios.shape == want.audios.shape == (2, 20480)
    assert rel_l2(torch.from_numpy(got.audios), torch.from_numpy(want.audios)) < 8e-2


def _tiny_pipe(seed=51):
    from audioldm_with_lora_amd.pipeline import AudioLDMPipeline
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    from audioldm_with_lora_amd.vae import AutoencoderKL
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    from oracle import configs
    torch.manual_seed(seed)
    return AudioLDMPipeline(AutoencoderKL(**configs.tiny_vae()), None, None, UNet2DConditionModel(**configs.tiny_unet()),
                            DDIMScheduler(), SpeechT5HifiGan(**configs.tiny_vocoder())).to("cuda")


def test_pipeline_call_surface_edge_cases():
    """The __call__ contract the reference scripts rely on [REF app.py:14, generate_audio.py:47-52]: waveform count and length,
    ragged lengths, CFG off, engine reuse / determinism, and the errors for bad arguments."""
    from audioldm_with_lora_amd._lib import AldmError
    pipe = _tiny_pipe()
    g = torch.Generator().manual_seed(1)
    pe = torch.nn.functional.normalize(torch.randn(2, 64, generator=g), dim=-1)
    # two waveforms per prompt -> 4 clips, trimmed to int(s * 16000); 1.3 s is not a whole number of latent rows
    out = pipe(prompt_embeds=pe, audio_length_in_s=1.3, num_inference_steps=3, num_waveforms_per_prompt=2,
               generator=torch.Generator().manual_seed(2))
    assert out.audios.shape == (4, int(1.3 * 16000)) and np.isfinite(out.audios).all()
    assert not np.allclose(out.audios[0], out.audios[1])                  # different noise per waveform
    # same seed -> bit-identical audio (engine and captured graph are reused); new prompt embeddings -> different audio
    a = pipe(prompt_embeds=pe, audio_length_in_s=1.3, num_inference_steps=3, num_waveforms_per_prompt=2, generator=torch.Generator().manual_seed(2))
    assert np.array_equal(a.audios, out.audios)
    b = pipe(prompt_embeds=-pe, audio_length_in_s=1.3, num_inference_steps=3, num_waveforms_per_prompt=2, generator=torch.Generator().manual_seed(2))
    assert not np.allclose(b.audios, out.audios)
    # guidance_scale <= 1: no CFG doubling; tuple return
    (wav,) = pipe(prompt_embeds=pe, audio_length_in_s=0.64, num_inference_steps=2, guidance_scale=1.0, return_dict=False)
    assert wav.shape == (2, 10240)
    t = pipe(prompt_embeds=pe, audio_length_in_s=0.64, num_inference_steps=2, output_type="pt").audios
    assert torch.is_tensor(t) and t.is_cuda and t.shape == (2, 10240)
    with pytest.raises(ValueError):
        pipe(audio_length_in_s=0.64)                                         # neither prompt nor prompt_embeds
    with pytest.raises(ValueError):
        pipe(prompt="a dog", audio_length_in_s=0.64)                         # no text encoder loaded
    with pytest.raises(ValueError):
        pipe(prompt_embeds=pe, audio_length_in_s=0.64, latents=torch.zeros(2, 8, 5, 16))
    with pytest.raises(NotImplementedError):
        pipe(prompt_embeds=pe, audio_length_in_s=0.64, eta=0.5)
    with pytest.raises(AldmError):
        pipe.to("cpu")(prompt_embeds=pe, audio_length_in_s=0.64)
