"""Oracle DDIM pins: closed-form known answers; integer arrays bit-exact."""
import os

import numpy as np
import torch

from oracle.ddim import DDIMScheduler

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ddim_tables.npz"))


def test_timesteps_bit_exact():
    s = DDIMScheduler()
    for n in (10, 50, 200):
        s.set_timesteps(n)
        assert s.timesteps.dtype == torch.int64
        assert np.array_equal(s.timesteps.numpy(), G[f"timesteps_{n}"])
        prev = np.array([s.prev_timestep(t) for t in s.timesteps])
        assert np.array_equal(prev, G[f"prev_{n}"])
    s.set_timesteps(10)
    assert s.timesteps.tolist() == [901, 801, 701, 601, 501, 401, 301, 201, 101, 1]
    s.set_timesteps(200)
    assert s.timesteps[:3].tolist() == [996, 991, 986] and int(s.timesteps[-1]) == 1


def test_alphas_cumprod_matches_float64():
    s = DDIMScheduler()
    assert s.alphas_cumprod.dtype == torch.float32
    np.testing.assert_allclose(s.alphas_cumprod.numpy(), G["alphas_cumprod_f64"], rtol=2e-5, atol=1e-6)
    assert float(s.final_alpha_cumprod) == float(s.alphas_cumprod[0])      # set_alpha_to_one = False


def test_step_eta0_closed_form_and_last_step():
    s = DDIMScheduler()
    s.set_timesteps(10)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 8, 5, 4, generator=g)
    e = torch.randn(2, 8, 5, 4, generator=g)
    ac = G["alphas_cumprod_f64"]
    for t in (901, 1):
        p = t - 100
        a, ap = ac[t], (ac[p] if p >= 0 else ac[0])
        x0 = (x.double() - (1 - a) ** 0.5 * e.double()) / a ** 0.5
        want = ap ** 0.5 * x0 + (1 - ap) ** 0.5 * e.double()
        got = s.step(e, t, x).prev_sample
        torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-5)


def test_add_noise():
    s = DDIMScheduler()
    g = torch.Generator().manual_seed(1)
    x, n = torch.randn(3, 8, 4, 4, generator=g), torch.randn(3, 8, 4, 4, generator=g)
    t = torch.tensor([0, 500, 999])
    got = s.add_noise(x, n, t)
    ac = torch.from_numpy(G["alphas_cumprod_f64"])[t].view(3, 1, 1, 1)
    want = ac.sqrt() * x.double() + (1 - ac).sqrt() * n.double()
    torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-5)
