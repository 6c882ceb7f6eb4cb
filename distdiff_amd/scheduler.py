"""Host-side DDIM schedule tables (what the reference gets from diffusers' DDIMScheduler.from_pretrained +
retrieve_timesteps, generate_data.py:863, 1043-1044). Only tables live here; the per-element update runs in the
cfg_ddim HIP kernel (elementwise.hip)."""
import numpy as np

from .config import SchedulerConfig


class DDIMSchedule:
    def __init__(self, cfg: SchedulerConfig = None):
        self.cfg = cfg or SchedulerConfig()
        c = self.cfg
        T = c.num_train_timesteps
        import torch  # same fp32 linspace / cumprod arithmetic as diffusers' DDIMScheduler.__init__ (host-side table only)
        if c.beta_schedule == "scaled_linear":
            betas = torch.linspace(c.beta_start ** 0.5, c.beta_end ** 0.5, T, dtype=torch.float32) ** 2
        elif c.beta_schedule == "linear":
            betas = torch.linspace(c.beta_start, c.beta_end, T, dtype=torch.float32)
        else:
            raise NotImplementedError(c.beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).numpy()
        self.final_alpha_cumprod = 1.0 if c.set_alpha_to_one else float(self.alphas_cumprod[0])
        self.timesteps = None

    def set_timesteps(self, n):
        c = self.cfg
        if c.timestep_spacing != "leading":
            raise NotImplementedError(c.timestep_spacing)
        ratio = c.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].astype(np.int64) + c.steps_offset
        self.timesteps = [int(t) for t in ts]
        return self.timesteps


def start_index(strength, n_steps):
    """generate_data.py:1174."""
    return int((1 - strength) * n_steps)


def guide_window(n_steps, guidance_step, guidance_period):
    """Index window of guide_timesteps = timesteps[n-guidance_step : n-guidance_step+guidance_period] (:1178)."""
    assert guidance_step >= 1
    first = n_steps - guidance_step
    assert 0 <= first and first + guidance_period <= n_steps
    return first, guidance_period
