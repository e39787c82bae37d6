"""Minimal DDIMScheduler with the behaviour the hot path consumes [3P diffusers==0.8.0]:
`timesteps[i]`, `add_noise(latent, noise, t)`  (optimize_token.py:25-34, ptp_utils.py:221-223)."""
from __future__ import annotations

import torch


class DDIMScheduler:
    def __init__(self, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", num_train_timesteps=1000,
                 clip_sample=True, set_alpha_to_one=True):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_train_timesteps = num_train_timesteps
        self.timesteps = torch.arange(0, num_train_timesteps).flip(0)
        self._cache = {}

    def set_timesteps(self, num_inference_steps: int):
        step = self.num_train_timesteps // num_inference_steps
        self.timesteps = (torch.arange(0, num_inference_steps) * step).flip(0)

    def add_noise(self, original_samples, noise, timesteps):
        t = int(timesteps) if not torch.is_tensor(timesteps) or timesteps.dim() == 0 else int(timesteps.reshape(-1)[0])
        acp = self.alphas_cumprod[t]
        a, b = float(acp.sqrt()), float((1 - acp).sqrt())
        return a * original_samples + b * noise
