"""Oracle: DDPM scheduler restatement (numpy/torch fp32, CPU).  TEST INFRASTRUCTURE ONLY.

The reference uses diffusers' `DDPMScheduler` loaded from the SD `scheduler_config.json`
(training/sid_sd_util.py:65) through exactly four members:
  add_noise(x0, noise, t)            sid_sd_util.py:182,191,242
  scale_model_input(x, t)            sid_sd_util.py:183,192,244,262   (identity for DDPM)
  step(eps, t, x_t).pred_original_sample   sid_sd_util.py:185,195,270
  config.prediction_type             sid_training_loop.py:424,438
diffusers is absent offline -> this restates the published DDPM formulas for the
SD config (beta_start 0.00085, beta_end 0.012, scaled_linear, 1000 steps, epsilon
prediction, clip_sample False); parity unpinned vs. the package, pinned numerically by
the closed-form constants in SURVEY.md section 8 (alpha_bar(625)=0.13776892 ...).

Deliberate difference, documented: diffusers' `step()` also draws (and discards, for our
purposes) variance noise when t>0.  We do not draw; "identical (z, t, noise)" is defined
by passing the tensors explicitly, never by seeding.
"""
from types import SimpleNamespace

import torch


class DDPMSchedulerRef:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, prediction_type='epsilon'):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.config = SimpleNamespace(prediction_type=prediction_type, num_train_timesteps=num_train_timesteps,
                                      beta_start=beta_start, beta_end=beta_end, beta_schedule='scaled_linear')

    def _coef(self, t, like):
        ac = self.alphas_cumprod.to(device=like.device, dtype=like.dtype)
        t = t.to(like.device)
        s0 = ac[t] ** 0.5
        s1 = (1 - ac[t]) ** 0.5
        s0 = s0.flatten()
        s1 = s1.flatten()
        while s0.ndim < like.ndim:
            s0 = s0.unsqueeze(-1)
            s1 = s1.unsqueeze(-1)
        return s0, s1

    def add_noise(self, x0, noise, timesteps):
        s0, s1 = self._coef(timesteps, x0)
        return s0 * x0 + s1 * noise

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict=True):
        t = timestep if torch.is_tensor(timestep) else torch.tensor(timestep)
        ac = self.alphas_cumprod.to(device=sample.device, dtype=sample.dtype)[t.to(sample.device)]
        x0 = (sample - (1 - ac) ** 0.5 * model_output) / ac ** 0.5
        return SimpleNamespace(pred_original_sample=x0)
