"""Model configurations of cvssp/audioldm-s-full-v2.

Written from the published JSON configs of the checkpoint the reference loads
[REF script/train/train_audioldm_lora.py:67,364-371]; no weights or JSONs are
available offline, so every value stays config-driven.
"""

UNET = dict(
    in_channels=8,
    out_channels=8,
    block_out_channels=(128, 256, 384, 640),
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    layers_per_block=2,
    num_heads=8,                      # diffusers' "attention_head_dim": 8 is the head COUNT
    cross_attention_dim=(128, 256, 384, 640),
    class_embed_input_dim=512,        # projection_class_embeddings_input_dim
    class_embeddings_concat=True,
    norm_num_groups=32,
    norm_eps=1e-5,
    flip_sin_to_cos=True,
    freq_shift=0,
    sample_size=128,                  # SURVEY.md A.1; only the pipeline's default clip length reads it
)

VAE = dict(
    in_channels=1,
    out_channels=1,
    latent_channels=8,
    block_out_channels=(128, 256, 512),
    layers_per_block=2,
    norm_num_groups=32,
    scaling_factor=0.9227914214134216,
)

SCHEDULER = dict(
    num_train_timesteps=1000,
    beta_start=0.0015,
    beta_end=0.0195,
    beta_schedule="scaled_linear",
    clip_sample=False,
    set_alpha_to_one=False,
    steps_offset=1,
    prediction_type="epsilon",
    timestep_spacing="leading",
)

VOCODER = dict(
    model_in_dim=64,
    sampling_rate=16000,
    upsample_initial_channel=1024,
    upsample_rates=(5, 4, 2, 2, 2),
    upsample_kernel_sizes=(16, 16, 8, 4, 4),
    resblock_kernel_sizes=(3, 7, 11),
    resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
    leaky_relu_slope=0.1,
    normalize_before=False,
)


# text_encoder/config.json of cvssp/audioldm-s-full-v2: ClapTextModelWithProjection (RoBERTa-base tower + 2-layer
# projection head).  [MEM]; the class itself is importable from the installed transformers, which pins oracle/clap_text.py.
CLAP_TEXT = dict(
    vocab_size=50265,
    hidden_size=768,
    num_hidden_layers=12,
    num_attention_heads=12,
    intermediate_size=3072,
    max_position_embeddings=514,
    type_vocab_size=1,
    pad_token_id=1,
    layer_norm_eps=1e-12,
    projection_dim=512,
)


def tiny_clap_text():
    return dict(CLAP_TEXT, vocab_size=200, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                intermediate_size=128, max_position_embeddings=40, projection_dim=32)


def tiny_unet():
    """A shrunken UNet of the same topology for fast CPU/GPU parity tests."""
    c = dict(UNET)
    c.update(block_out_channels=(32, 64, 96, 160), cross_attention_dim=(32, 64, 96, 160),
             class_embed_input_dim=64, norm_num_groups=8, num_heads=4)   # head dims 16/24/40
    return c


def tiny_vae():
    c = dict(VAE)
    c.update(block_out_channels=(32, 64, 128), norm_num_groups=8)
    return c


def tiny_vocoder():
    c = dict(VOCODER)
    c.update(upsample_initial_channel=128)
    return c
