"""CPU oracle for the AudioLDM+LoRA hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain torch-CPU fp32 *restatement* of the arithmetic the
reference (2025-comprehensive-design/AudioLDM-with-LoRA) reaches through its
un-vendored pip dependencies: diffusers==0.32.2 (AudioLDMPipeline,
UNet2DConditionModel, DDIMScheduler, AutoencoderKL), peft==0.13.2
(lora.Linear) and transformers==4.29.0 (SpeechT5HifiGan,
ClapTextModelWithProjection) --
[REF requirements.txt:24,90,149].  The reference's own call sites are
[REF script/train/train_audioldm_lora.py:364-371,378-385,495-565],
[REF script/inference/generate_audio.py:18-52] and [REF app.py:7-14].

PARITY PIN STATUS
  * vocoder (hifigan.py): PINNED against the importable
    transformers.SpeechT5HifiGan class (tests/test_oracle_vocoder.py and
    tests/golden/vocoder_*.npz, made by tests/golden/make_golden.py).
  * CLAP text tower (clap_text.py): PINNED against the importable
    transformers.ClapTextModelWithProjection (tests/test_oracle_clap_text.py,
    tests/golden/clap_text_tiny.npz).
  * log-mel front end (mel.py): the STFT is torch.stft -- the call the reference
    itself makes [REF script/data/datasets.py:327-338] -- and the Slaney mel
    basis is PINNED against transformers.audio_utils.mel_filter_bank
    (tests/test_oracle_mel.py); librosa is not installed.
  * building blocks (conv / group-norm / SDPA / layer-norm / GEGLU /
    interpolate / AdamW / polynomial LR): pinned against torch / transformers
    primitives, the same primitives diffusers composes.
  * whole-model UNet / VAE / DDIM / peft-LoRA graph wiring: **parity unpinned**
    -- diffusers and peft are not installed in the build image, there is no
    network and the reference ships no tests or golden vectors (SURVEY.md
    section 8c).  The graph is restated from the published diffusers/peft
    source layout; structural known-answers (param counts, key/shape
    manifest, DDIM closed forms) are the only anchors.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product path (audioldm_with_lora_amd) never does.
"""
