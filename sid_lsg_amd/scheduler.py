"""DDPM scheduler with the four members the reference touches (training/sid_sd_util.py:182-185,
191-195, 242-244, 262, 270; training/sid_training_loop.py:424,438): add_noise, scale_model_input,
step(...).pred_original_sample, config.prediction_type.  SD `scheduler_config.json` values:
scaled_linear betas 0.00085..0.012, 1000 steps, epsilon prediction, no sample clipping.

`coefficients(t)` returns the per-sample (sqrt(abar_t), sqrt(1-abar_t)) pair that the fused HIP glue
kernels (sidlsg_noisy_input / sidlsg_cfg_x0) consume, with no host synchronisation (the reference's
per-sample `scheduler.step` loop, sid_sd_util.py:270, costs 2*b host syncs per call).
"""
from types import SimpleNamespace

import torch


class DDPMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, prediction_type='epsilon'):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self._s0 = self.alphas_cumprod ** 0.5
        self._s1 = (1 - self.alphas_cumprod) ** 0.5
        self.config = SimpleNamespace(prediction_type=prediction_type, num_train_timesteps=num_train_timesteps,
                                      beta_start=beta_start, beta_end=beta_end, beta_schedule='scaled_linear',
                                      clip_sample=False)

    def to(self, device):
        for k in ('betas', 'alphas_cumprod', '_s0', '_s1'):
            setattr(self, k, getattr(self, k).to(device))
        return self

    def __repr__(self):
        return f'DDPMScheduler(scaled_linear, {self.config.num_train_timesteps} steps, {self.config.prediction_type})'

    def coefficients(self, t):
        if self._s0.device != t.device:
            self.to(t.device)
        t = t.reshape(-1)
        return self._s0[t].contiguous(), self._s1[t].contiguous()

    # ---- generic duck-typed API (plain tensor math; used with non-HIP networks and on the cold path)
    def add_noise(self, original_samples, noise, timesteps):
        s0, s1 = self.coefficients(timesteps.to(original_samples.device))
        shape = (-1,) + (1,) * (original_samples.ndim - 1)
        return s0.to(original_samples.dtype).view(shape) * original_samples + s1.to(noise.dtype).view(shape) * noise

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict=True):
        t = timestep if torch.is_tensor(timestep) else torch.tensor(timestep)
        s0, s1 = self.coefficients(t.to(sample.device))
        s0, s1 = s0.to(sample.dtype), s1.to(sample.dtype)
        if s0.numel() > 1:
            shape = (-1,) + (1,) * (sample.ndim - 1)
            s0, s1 = s0.view(shape), s1.view(shape)
        return SimpleNamespace(pred_original_sample=(sample - s1 * model_output) / s0)

    def get_velocity(self, sample, noise, timesteps):
        s0, s1 = self.coefficients(timesteps.to(sample.device))
        shape = (-1,) + (1,) * (sample.ndim - 1)
        return s0.view(shape) * noise - s1.view(shape) * sample
