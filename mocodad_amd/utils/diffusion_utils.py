"""Drop-in for the reference's utils/diffusion_utils.py (API surface kept: `betas_for_alpha_bar`,
`Diffusion` with .beta/.alpha/.alpha_hat, schedule_noise, prepare_noise_schedule, noise_images/
noise_graph/noise_latent, sample_timesteps — /root/reference/utils/diffusion_utils.py:8-75).

The eval hot path only needs the tables; `step_table` additionally lays them out the way the HIP
kernel consumes them (one row per reverse step: update coefficients + sinusoidal embedding)."""
import math
from typing import Tuple

import numpy as np
import torch


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    t = np.arange(num_diffusion_timesteps + 1, dtype=np.float64) / num_diffusion_timesteps
    ab = np.array([alpha_bar(v) for v in t])
    return np.minimum(1.0 - ab[1:] / ab[:-1], max_beta)


def _cosine_alpha_bar(t):
    return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2


class Diffusion:
    def __init__(self, noise_steps=50, beta_start=1e-4, beta_end=0.02, device="cuda", time=25, n_joints=22):
        self.noise_steps = noise_steps
        self.beta_start = beta_start
        self.beta_end = beta_end
        self.time = time
        self.joints = n_joints
        self.device = device
        self.beta = self.schedule_noise()
        self.alpha = 1.0 - self.beta
        self.alpha_hat = torch.cumprod(self.alpha, dim=0)

    def prepare_noise_schedule(self) -> torch.Tensor:
        return torch.linspace(self.beta_start, self.beta_end, self.noise_steps, device=self.device)

    def schedule_noise(self) -> torch.Tensor:
        betas = betas_for_alpha_bar(self.noise_steps, _cosine_alpha_bar)
        return torch.tensor(betas, dtype=torch.float32, device=self.device)

    def _forward_noise(self, x, t, nd):
        ah = self.alpha_hat.to(t.device)[t]
        shape = (-1,) + (1,) * nd
        eps = torch.randn_like(x)
        return torch.sqrt(ah).view(shape) * x + torch.sqrt(1 - ah).view(shape) * eps, eps

    def noise_images(self, x: torch.Tensor, t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._forward_noise(x, t, 3)

    def noise_graph(self, x: torch.Tensor, t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._forward_noise(x, t, 3)

    def noise_latent(self, x: torch.Tensor, t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._forward_noise(x, t, 1)

    def sample_timesteps(self, n: int) -> torch.Tensor:
        return torch.randint(low=1, high=self.noise_steps, size=(n,))


def pos_encoding(t: torch.Tensor, channels: int) -> torch.Tensor:
    """Sinusoidal step embedding of STSE_Unet.pos_encoding (stsae_unet.py:161-179); t: (N,1) float."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, channels, 2).float() / channels))
    a = t.repeat(1, channels // 2) * inv_freq
    return torch.cat([torch.sin(a), torch.cos(a)], dim=-1)


def step_table(noise_steps: int, emb_dim: int = 16) -> torch.Tensor:
    """(ns, 4+emb_dim) fp32 CPU table: row i = [1/sqrt(alpha_i), (1-alpha_i)/sqrt(1-alpha_hat_i), sqrt(beta_i), 0,
    pos_encoding(i)] — the scalars of the update at models/mocodad.py:172-178, computed with the same
    fp32 torch ops the reference uses so the coefficients are bit-identical."""
    d = Diffusion(noise_steps=noise_steps, device="cpu")
    beta, alpha, ah = d.beta, d.alpha, d.alpha_hat
    tab = torch.zeros(noise_steps, 4 + emb_dim, dtype=torch.float32)
    tab[:, 0] = 1 / torch.sqrt(alpha)
    tab[:, 1] = (1 - alpha) / torch.sqrt(1 - ah)
    tab[:, 2] = torch.sqrt(beta)
    tab[:, 4:] = pos_encoding(torch.arange(noise_steps, dtype=torch.float32)[:, None], emb_dim)
    return tab
