"""Oracle: noise schedule, timestep spacing and the per-step sampler algebra (src/schedulers.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain PyTorch-CPU, f32 exactly where the
reference is f32 (and f64 where its numpy detours make it f64).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

SAMPLERS = ("ddpm", "ddim", "ddim_simple", "ddim_orig", "ddim_simple_orig", "ddim_simple_drag", "ddpm_orig")


def interp1d(x: torch.Tensor, y: torch.Tensor, xnew: torch.Tensor) -> torch.Tensor:
    """Interp1d.forward for 1-D x, y, xnew (src/torchinterp1d.py:10-154): returns shape (1, P)."""
    eps = torch.finfo(y.dtype).eps
    xn = xnew[None, :]
    ind = torch.searchsorted(x.contiguous(), xn.contiguous()) - 1
    ind = torch.clamp(ind, 0, x.shape[0] - 2)
    slopes = (y[1:] - y[:-1]) / (eps + (x[1:] - x[:-1]))
    return y[ind] + slopes[ind] * (xn - x[ind])


def space_timesteps(num_timesteps: int, count: int):
    """space_timesteps with a single integer section (src/schedulers.py:38-91)."""
    if num_timesteps < count:
        raise ValueError(f"cannot divide section of {num_timesteps} steps into {count}")
    stride = 1 if count <= 1 else (num_timesteps - 1) / (count - 1)
    cur, taken = 0.0, []
    for _ in range(count):
        taken.append(round(cur))
        cur += stride
    return set(taken)


def replace_duplicate_t(ts: torch.Tensor, max_step: int = 999) -> torch.Tensor:
    """src/schedulers.py:15-31: make the timestep list strictly decreasing and <= max_step."""
    a = torch.zeros_like(ts)
    a[-2:] = ts[-2:]
    for i in range(len(ts) - 1, 0, -1):
        a[i - 1] = ts[i - 1] if ts[i - 1] > a[i] else a[i] + 1
    b = torch.zeros_like(a)
    big = max_step
    for i in range(len(a) - 1):
        b[i] = big if a[i] > big else a[i]
        big = b[i] - 1
    return b


@dataclass
class Schedule:
    name: str
    num_train_timesteps: int
    betas: torch.Tensor
    alphas_cumprod: torch.Tensor
    sigmas: torch.Tensor                    # ascending table, sqrt(1/abar - 1)
    final_sigma: torch.Tensor
    final_alpha_cumprod: torch.Tensor
    train_timesteps: torch.Tensor
    sampler_var: str
    eta: float
    set_alpha_to_one: bool
    continuous_t: bool = False
    timesteps: Optional[torch.Tensor] = None
    sampling_sigmas: Optional[torch.Tensor] = None
    min_var_coef: Optional[torch.Tensor] = None
    num_inference_steps: int = 0

    # ---- t <-> sigma maps (src/schedulers.py:185-220,306-348)
    def sigma_to_t(self, sigma):
        return torch.searchsorted(self.sigmas, sigma)

    def t_to_sigma_interp(self, t):
        xnew = t.squeeze()
        if xnew.dim() == 0:
            xnew = xnew.unsqueeze(0)
        y_new = interp1d(self.train_timesteps.float(), self.alphas_cumprod, xnew).squeeze(0)
        sigma = (1 / y_new - 1).sqrt()
        return torch.where(t >= 0, sigma, self.final_sigma).float()

    def sigma_to_t_interp(self, sigma):
        xnew = sigma.squeeze()
        if xnew.dim() == 0:
            xnew = xnew.unsqueeze(0)
        return interp1d(self.sigmas, self.train_timesteps.float(), xnew).squeeze(0).float()

    def get_sigma(self, t):
        if self.continuous_t:
            return self.t_to_sigma_interp(t)
        return torch.where(t >= 0, self.sigmas[t], self.final_sigma)

    def get_alpha_bar(self, t):
        if self.continuous_t:
            return 1 / (self.t_to_sigma_interp(t) ** 2 + 1)
        return torch.where(t >= 0, self.alphas_cumprod[t], self.final_alpha_cumprod)

    def get_t_from_sigma(self, sigma):
        return self.sigma_to_t_interp(sigma) if self.continuous_t else self.sigma_to_t(sigma)

    # ---- log-variance (src/schedulers.py:367-390)
    def get_eps_logvar(self, sigma_t, sigma_prev, learned_logvar=None):
        beta_t = ((sigma_t ** 2 - sigma_prev ** 2) / (sigma_t ** 2 + 1)).abs().clamp(min=1e-20)
        alpha_t = 1 / (sigma_t ** 2 + 1)
        alpha_prev = 1 / (sigma_prev ** 2 + 1)
        coef = ((1 - alpha_prev) / (1 - alpha_t)).clamp(min=0, max=1)
        max_logvar = beta_t.log()
        min_logvar = (beta_t * coef).clamp(min=self.min_var_coef).log()
        if self.sampler_var == "learned":
            frac = (learned_logvar + 1) / 2
            return frac * max_logvar + (1 - frac) * min_logvar
        if self.sampler_var == "fixedsmall":
            return min_logvar
        if self.sampler_var == "fixedlarge":
            return max_logvar
        return None

    def pred_xstart(self, xt, eps, sigma_t):
        """src/schedulers.py:407-409"""
        return xt - sigma_t * eps

    def pred_xprev(self, x0, eps, sigma_t, sigma_prev, xt=None, log_variance=None, noise=None):
        """pred_xprev of the seven live sampler classes (src/schedulers.py:432-449,465-473,487-496,505-514,
        548-562,581-599,609-627).  ``noise`` defaults to torch.randn_like(x0) drawn from the global
        generator exactly where the reference draws it."""
        def draw():
            return torch.randn_like(x0) if noise is None else noise
        eta, name = self.eta, self.name
        if name == "ddim":
            if eta > 0:
                noise_sigma = eta * torch.exp(0.5 * log_variance) / torch.sqrt(1 / (sigma_prev ** 2 + 1))
                z = (sigma_prev > 0) * draw()
            else:
                noise_sigma, z = 0, 0
            signal_sigma = torch.sqrt((sigma_prev ** 2 - noise_sigma ** 2).clamp(min=0))
            noise_sigma = torch.sqrt(sigma_prev ** 2 - signal_sigma ** 2)
            return x0 + signal_sigma * eps + noise_sigma * z
        if name in ("ddim_simple", "ddim_simple_orig", "ddim_simple_drag"):
            if name != "ddim_simple":
                eps = (xt - x0) / sigma_t
            signal_sigma = sigma_prev if name == "ddim_simple_drag" else math.sqrt(1 - eta ** 2) * sigma_prev
            x_prev = x0 + signal_sigma * eps
            if eta > 0:
                x_prev = x_prev + (eta * sigma_prev) * draw()
            return x_prev
        if name == "ddpm":
            noise_sigma = torch.exp(0.5 * log_variance) / torch.sqrt(1 / (sigma_prev ** 2 + 1))
            signal_sigma = torch.sqrt((sigma_prev ** 2 - noise_sigma ** 2).clamp(min=0))
            x_prev = x0 + signal_sigma * eps
            return x_prev + noise_sigma * ((sigma_prev > 0) * draw())
        if name == "ddim_orig":
            eps = (xt - x0) / sigma_t
            if eta > 0:
                noise_sigma = eta * torch.exp(0.5 * log_variance) / torch.sqrt(1 / (sigma_prev ** 2 + 1))
                z = (sigma_prev > 0).float() * draw()
            else:
                noise_sigma, z = 0.0, 0.0
            signal_sigma = torch.sqrt((sigma_prev ** 2 - noise_sigma ** 2).clamp(min=0))
            return x0 + signal_sigma * eps + noise_sigma * z
        if name == "ddpm_orig":
            alpha_bar = 1 / (sigma_t ** 2 + 1)
            alpha_bar_prev = 1 / (sigma_prev ** 2 + 1)
            alpha_t = alpha_bar / alpha_bar_prev
            beta_t = 1 - alpha_t
            zt = xt * alpha_bar.sqrt()
            c1 = beta_t * alpha_bar_prev.sqrt() / (1.0 - alpha_bar)
            c2 = (1.0 - alpha_bar_prev) * alpha_t.sqrt() / (1.0 - alpha_bar)
            mean = c1 * x0 + c2 * zt
            z_prev = mean + (sigma_prev > 0).float() * torch.exp(0.5 * log_variance) * draw()
            return z_prev / alpha_bar_prev.sqrt()
        raise NotImplementedError(name)


def make_betas(num_train_timesteps, beta_start, beta_end, beta_schedule):
    """Scheduler.__init__ (src/schedulers.py:106-127)."""
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "quadratic":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    if beta_schedule == "cosine":
        s = 0.008
        x = torch.linspace(0, num_train_timesteps, num_train_timesteps + 1)
        ac = torch.cos(((x / num_train_timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
        ac = ac / ac[0]
        return torch.clip(1 - (ac[1:] / ac[:-1]), 1e-6, 0.999)
    if beta_schedule == "sigmoid":
        return torch.sigmoid(torch.linspace(-6, 6, num_train_timesteps)) * (beta_end - beta_start) + beta_start
    raise NotImplementedError(beta_schedule)


def get_sampler(sampler_name, train_timesteps, inference_timesteps, beta_start=0.0001, beta_end=0.02,
                beta_schedule="linear", sigma_style="DDIM", set_alpha_to_one=True, start_sigma=None, end_sigma=None,
                sampler_var="none", continuous_t=False, linear_scale=1.0, eta=0.0, start_t=None, end_t=None) -> Schedule:
    """get_sampler + Scheduler.__init__ + set_timesteps_sigma (src/schedulers.py:96-164,227-284,676-726)."""
    if sampler_name not in SAMPLERS:
        raise NotImplementedError(sampler_name)
    if sampler_name == "ddpm_orig":
        eta = 1.0                                            # :577
    betas = make_betas(train_timesteps, beta_start, beta_end, beta_schedule)
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    final_ac = torch.tensor(1.0)                             # always 1 (:133)
    S = Schedule(name=sampler_name, num_train_timesteps=train_timesteps, betas=betas, alphas_cumprod=alphas_cumprod,
                 sigmas=(1 / alphas_cumprod - 1).sqrt(), final_sigma=(1 / final_ac - 1).sqrt(),
                 final_alpha_cumprod=final_ac,
                 train_timesteps=torch.tensor(np.arange(0, train_timesteps).astype(np.int64)),
                 sampler_var=sampler_var, eta=eta, set_alpha_to_one=bool(set_alpha_to_one))
    # start / end sigma defaults (:711-723)
    if start_sigma is None or start_sigma <= 0:
        start = S.sigmas[-1] if (start_t is None or start_t < 0) else min(S.sigmas[start_t], S.sigmas[-1])
    else:
        start = torch.tensor(min(start_sigma, S.sigmas[-1]))
    if end_sigma is None or end_sigma <= 0:
        end = S.sigmas[0] if (end_t is None or end_t < 0) else S.sigmas[end_t]
    else:
        end = end_sigma
    set_timesteps_sigma(S, start, end, inference_timesteps, sigma_style, linear_scale, continuous_t)
    return S


def set_timesteps_sigma(S: Schedule, start, end, num_inference_steps, style="DDIM", scale=1.0, continuous_t=False):
    """src/schedulers.py:227-284"""
    S.continuous_t = bool(continuous_t)
    S.num_inference_steps = num_inference_steps
    dtype = torch.long if not continuous_t else torch.float32
    n = num_inference_steps if S.set_alpha_to_one else num_inference_steps + 1
    if style == "DDIM":
        start_t = S.get_t_from_sigma(torch.as_tensor(start)).item()
        end_t = S.get_t_from_sigma(torch.as_tensor(end)).item()
        ts = space_timesteps(start_t + 1 - end_t, n)
        ts = end_t + np.array(sorted(ts, reverse=True))
        S.timesteps = torch.tensor(ts, dtype=dtype)
        sigmas = S.get_sigma(S.timesteps)
    elif style == "EDM":
        rho = 7
        sigmas = torch.tensor([(start ** (1 / rho) + i / (n - 1) * (end ** (1 / rho) - start ** (1 / rho))) ** rho
                               for i in range(n)])
        S.timesteps = S.get_t_from_sigma(sigmas)
    elif style == "Linear":
        sigmas = torch.tensor(np.exp(np.linspace(np.log(start), np.log(end), n)))
        S.timesteps = S.get_t_from_sigma(sigmas)
    elif style == "Scaled":
        diff = np.log(end) - np.log(start)
        a_t = scale ** np.arange(n - 1)
        cs = np.cumsum(a_t)
        logs = np.insert(np.log(start) + diff / cs[-1] * cs, 0, np.log(start))
        sigmas = torch.tensor(np.exp(logs))
        S.timesteps = S.get_t_from_sigma(sigmas)
    else:
        raise ValueError("Invalid style!")
    S.timesteps = S.timesteps.squeeze()
    sigmas = sigmas.squeeze()
    if not continuous_t:
        S.timesteps = replace_duplicate_t(S.timesteps)
        S.sampling_sigmas = S.get_sigma(S.timesteps)
    else:
        S.sampling_sigmas = sigmas
    if S.set_alpha_to_one:
        S.timesteps = torch.cat([S.timesteps, torch.tensor([-1])])
        S.sampling_sigmas = torch.cat([S.sampling_sigmas, torch.tensor([S.final_sigma])])
    st, sp = S.sampling_sigmas[-3], S.sampling_sigmas[-2]
    beta_t = (st ** 2 - sp ** 2) / (st ** 2 + 1)
    S.min_var_coef = beta_t * (1 - 1 / (sp ** 2 + 1)) / (1 - 1 / (st ** 2 + 1))
