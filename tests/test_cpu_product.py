"""CPU-side checks of the product: the C-ABI library loads and exports every declared symbol, host structures match
the header, packing / manifest / scheduler / LoRA host logic, and the no-fallback rule."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from audioldm_with_lora_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def test_library_exports_every_symbol_in_header(lib):
    hdr = open(os.path.join(ROOT, "include", "aldm_hip.h")).read()
    declared = set(re.findall(r"\b(aldm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"aldm_igemm_t"}
    assert declared == set(lib.PROTOTYPES), declared ^ set(lib.PROTOTYPES)
    l = lib.load()
    for name in declared:
        assert hasattr(l, name)
    assert l.aldm_version().startswith(b"aldm_hip")


def test_igemm_struct_layout_matches_header(lib, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "aldm_hip.h"\nint main(){printf("%zu %zu %zu %zu\\n",'
                   'sizeof(aldm_igemm_t),offsetof(aldm_igemm_t,w),offsetof(aldm_igemm_t,out),offsetof(aldm_igemm_t,workspace));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    size, off_w, off_out, off_ws = map(int, subprocess.check_output([str(exe)]).split())
    A = lib.IgemmArgs
    assert (ctypes.sizeof(A), A.w.offset, A.out.offset, A.workspace.offset) == (size, off_w, off_out, off_ws)


def test_bad_arguments_are_rejected_without_a_gpu(lib):
    l = lib.load()
    a = lib.IgemmArgs()
    assert l.aldm_igemm(ctypes.byref(a), None) == -1          # ALDM_E_ARG: null pointers
    assert b"igemm" in l.aldm_last_error()
    assert l.aldm_layernorm(None, 0, 0, None, None, 1e-5, None, None) == -1


def test_no_cpu_fallback():
    from audioldm_with_lora_amd import ops
    from audioldm_with_lora_amd._lib import AldmError
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    from oracle import configs
    u = UNet2DConditionModel(**configs.tiny_unet())
    with pytest.raises(AldmError):
        u(torch.zeros(1, 8, 8, 8), 1, class_labels=torch.zeros(1, 64))
    with pytest.raises(AldmError):
        ops.conv(torch.zeros(1, 4, 4, 8, dtype=torch.bfloat16), ops.pack_conv(torch.zeros(8, 8, 3, 3), None))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "audioldm_with_lora_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_configs_and_state_dict_manifests_equal_oracle():
    from audioldm_with_lora_amd import configs as pc
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    from audioldm_with_lora_amd.vae import AutoencoderKL
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    from oracle import configs as oc
    from oracle.hifigan import SpeechT5HifiGan as OVoc
    from oracle.unet import UNet2DConditionModel as OUNet
    from oracle.vae import AutoencoderKL as OVae
    for name in ("UNET", "VAE", "SCHEDULER", "VOCODER"):
        assert getattr(pc, name) == getattr(oc, name)
    for mine, ref in ((UNet2DConditionModel(), OUNet()), (AutoencoderKL(), OVae()), (SpeechT5HifiGan(), OVoc())):
        a = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
        b = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        assert a == b


def test_scheduler_integer_indexing_bit_exact():
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    G = np.load(os.path.join(ROOT, "tests", "golden", "ddim_tables.npz"))
    s = DDIMScheduler()
    for n in (10, 50, 200):
        s.set_timesteps(n)
        assert s.timesteps.dtype == torch.int64
        assert np.array_equal(s.timesteps.numpy(), G[f"timesteps_{n}"])
        assert np.array_equal(np.array([s.prev_timestep(t) for t in s.timesteps]), G[f"prev_{n}"])
        tab = s.coefficient_table()
        assert tab.shape == (n, 4) and tab.dtype == torch.float32
        ac = G["alphas_cumprod_f64"]
        t_last = int(s.timesteps[-1])
        np.testing.assert_allclose(tab[-1, 2].item() ** 2, ac[0], rtol=1e-5)          # final step uses alpha_bar[0]
        np.testing.assert_allclose(tab[-1, 0].item() ** 2, ac[t_last], rtol=1e-5)


def test_lora_helpers_peft_shape():
    from audioldm_with_lora_amd import lora as L
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    from oracle import configs
    u = UNet2DConditionModel(**configs.tiny_unet())
    pm = L.get_peft_model(u, L.LoraConfig(r=2, lora_alpha=2, target_modules=["to_q", "to_v"], init_lora_weights="gaussian"))
    assert sum(isinstance(m, L.LoraLinear) for m in u.modules()) == 64
    tr = [n for n, p in pm.named_parameters() if p.requires_grad]
    assert len(tr) == 128 and all("lora_" in n for n in tr)
    k = "base_model.model.down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.lora_A.default.weight"
    assert k in pm.state_dict()
    sd = L.get_peft_model_state_dict(pm)
    assert len(sd) == 128 and k.replace(".default", "") in sd
    assert any(x.endswith("to_q.lora.down.weight") for x in L.convert_state_dict_to_diffusers(sd))
    b = [p for n, p in pm.named_parameters() if "lora_B" in n]
    assert all(float(p.abs().max()) == 0.0 for p in b)


def test_packing_layouts():
    from audioldm_with_lora_amd import ops
    w = torch.arange(2 * 16 * 3 * 3, dtype=torch.float32).view(2, 16, 3, 3)
    pw = ops.pack_conv(w, None)
    assert pw.w.shape == (2, 192) and pw.Kpad % 64 == 0
    assert float(pw.w[1, (1 * 3 + 2) * 16 + 5]) == float(w[1, 5, 1, 2].to(torch.bfloat16))     # K = (kh, kw, cin)
    g = ops.pack_geglu(torch.arange(64 * 8, dtype=torch.float32).view(64, 8), torch.arange(64, dtype=torch.float32))
    assert g.bias[:16].tolist() == list(range(16)) and g.bias[16:32].tolist() == list(range(32, 48))   # 16 value | 16 gate
    lin = ops.pack_linear(torch.zeros(32, 64), None)
    ops.attach_lora(lin, [(0, 16, torch.ones(4, 64), torch.ones(16, 4), 2.0), (16, 16, torch.ones(8, 64), torch.ones(16, 8), 1.0)])
    assert lin.Rp == 32 and lin.lora_a.shape == (32, 64) and lin.lora_b.shape == (32, 32)
    assert float(lin.lora_b[0, 0]) == 2.0 and float(lin.lora_b[0, 4]) == 0.0 and float(lin.lora_b[16, 4]) == 1.0
    assert ops.pick_tile(32000, 128) == 3 and ops.pick_tile(512, 640) == 2 and ops.pick_tile(32000, 256) == 6
    assert ops.auto_splits(512, 640, 90) == 4 and ops.auto_splits(32000, 128, 18) == 1
    # ring depth: 500 workgroups of the 128x64 tile must stay 2-per-CU resident (3 stages), 80x4 split-K ones take the deepest ring
    assert ops.pick_ring(3, 128, 64, 0, 500, 18) == 3 and ops.pick_ring(2, 64, 64, 0, 320, 90) == 4 and ops.pick_ring(2, 64, 64, 32, 1500, 4) == 2


def test_vocoder_transposed_conv_phase_decomposition_matches_torch():
    """Host-side phase math (which taps / which output rows) checked with plain torch on the CPU."""
    from audioldm_with_lora_amd import vocoder as V
    g = torch.Generator().manual_seed(0)
    for (k, u) in ((16, 5), (16, 4), (8, 2), (4, 2)):
        p = (k - u) // 2
        w = torch.randn(8, 8, k, generator=g)
        x = torch.randn(1, 8, 9, generator=g)
        want = torch.nn.functional.conv_transpose1d(x, w, None, stride=u, padding=p)
        out_len = want.shape[2]
        got = torch.zeros_like(want)
        ntap = (k + u - 1) // u
        for phi in range(u):
            q0 = max(0, -((phi - p) // u))
            t0 = u * q0 + phi - p
            if t0 >= out_len:
                continue
            nq = (out_len - 1 - t0) // u + 1
            for qi in range(nq):
                q = qi + q0
                acc = torch.zeros(8)
                for i, j in enumerate(range(phi, k, u)):
                    s = q - i
                    if 0 <= s < x.shape[2]:
                        acc += w[:, :, j].t() @ x[0, :, s]
                got[0, :, t0 + u * qi] = acc
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)


def test_tuned_table_is_well_formed():
    """tuned_gfx950.json (tools/autotune.py): every entry is a launchable (tile, ring, splits) triple."""
    import json
    import os
    from audioldm_with_lora_amd import ops
    assert os.path.exists(ops.TUNED_PATH)
    table = json.load(open(ops.TUNED_PATH))["igemm"]
    assert table, "empty table"
    for key, (tile, ring, splits) in table.items():
        assert key.startswith("M") and tile in ops.TILE_DIMS and 2 <= ring <= 4 and 1 <= splits <= 16, (key, tile, ring, splits)
        if " vt1 " in key or " ln1 " in key:
            assert splits == 1
    assert ops.TUNED == {k: tuple(v) for k, v in table.items()} or os.environ.get("ALDM_NO_TUNED") == "1"


def test_collate_contract_and_caption_lengths_host_logic():
    """the trainer's input batch has the reference collate_fn's keys / shapes / dtypes
    [REF script/train/train_audioldm_lora.py:415-420]; caption lengths come from a right-padded mask only."""
    import pytest
    import torch
    from audioldm_with_lora_amd.clap_text import ClapTextModelWithProjection
    from audioldm_with_lora_amd.script.train import synthetic_batch
    b = synthetic_batch(3, torch.Generator().manual_seed(0))
    assert b["log_mel_spec"].shape == (3, 1, 1024, 64) and b["log_mel_spec"].dtype == torch.float32
    assert b["input_ids"].shape == (3, 1, 512) and b["input_ids"].dtype == torch.int64
    assert b["attention_mask"].shape == (3, 1, 512)
    ids, mask = b["input_ids"].squeeze(1), b["attention_mask"].squeeze(1)
    lens = ClapTextModelWithProjection._lengths(ids, mask)
    assert lens.dtype == torch.int32 and lens.tolist() == mask.sum(1).tolist() and 8 <= int(lens.min()) and int(lens.max()) <= 64
    assert bool((ids[mask == 0] == 1).all()) and bool((ids[:, 0] == 0).all())          # <pad> = 1 after the caption, <s> = 0 first
    assert ClapTextModelWithProjection._lengths(ids, None).tolist() == [512] * 3
    bad = mask.clone()
    bad[0, 0] = 0                                                                          # left padding
    with pytest.raises(ValueError):
        ClapTextModelWithProjection._lengths(ids, bad)
    with pytest.raises(ValueError):
        ClapTextModelWithProjection._lengths(ids, torch.zeros_like(mask))                  # an empty caption


def test_clap_text_manifest_equals_oracle_and_no_cpu_forward():
    import pytest
    import torch
    from audioldm_with_lora_amd._lib import AldmError
    from audioldm_with_lora_amd.clap_text import ClapTextModelWithProjection
    from audioldm_with_lora_amd.configs import tiny_clap_text
    from oracle.clap_text import ClapTextModelWithProjection as OClap
    m, o = ClapTextModelWithProjection(**tiny_clap_text()), OClap(**tiny_clap_text())
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in o.state_dict().items()}
    assert any(k == "text_model.encoder.layer.0.attention.self.query.weight" for k in m.state_dict())
    with pytest.raises(AldmError):
        m(input_ids=torch.zeros(1, 8, dtype=torch.long))
