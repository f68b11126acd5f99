"""Noise schedules and samplers with the reference's class API (drop-in for src/schedulers.py).

Host side (init only, torch CPU tensors as in the reference): beta / alpha-bar / sigma tables,
DDIM timestep spacing, the sigma<->t maps.  Device side: the per-step algebra of every sampler is
the fused HIP kernel ``nlc_sched_step`` (see csrc/sampler.hip); ``pred_xstart`` / ``pred_xprev``
here launch it on GPU tensors so the reference's call sequence
(src/experiments.py:360-370) keeps working unchanged.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np
import torch

from . import ops
from ._ext import CLIP_MODES, SCHED_VARIANTS, VAR_MODES, NlcError, SchedDesc


def space_timesteps(num_timesteps, section_counts):
    """src/schedulers.py:38-91 (string 'ddimN' striding and comma-separated section counts)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += frac
        start += size
    return set(steps)


def replace_duplicate_t(ts, max_step=999):
    """src/schedulers.py:15-31"""
    up = torch.zeros_like(ts)
    up[-2:] = ts[-2:]
    for i in range(len(ts) - 1, 0, -1):
        up[i - 1] = ts[i - 1] if ts[i - 1] > up[i] else up[i] + 1
    capped = torch.zeros_like(up)
    ceiling = max_step
    for i in range(len(up) - 1):
        capped[i] = ceiling if up[i] > ceiling else up[i]
        ceiling = capped[i] - 1
    return capped


def _interp1d(x, y, xnew):
    """Interp1d.forward for 1-D inputs (src/torchinterp1d.py:10-154)."""
    eps = torch.finfo(y.dtype).eps
    xn = xnew[None, :]
    ind = torch.clamp(torch.searchsorted(x.contiguous(), xn.contiguous()) - 1, 0, x.shape[0] - 2)
    slopes = (y[1:] - y[:-1]) / (eps + (x[1:] - x[:-1]))
    return y[ind] + slopes[ind] * (xn - x[ind])


class Scheduler:
    """src/schedulers.py:95-423.  ``variant`` selects the pred_xprev formula of the subclass."""
    variant = None

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 set_alpha_to_one=True, sampler_var="none", eta=0.0):
        n = num_train_timesteps
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
        elif beta_schedule == "quadratic":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
        elif beta_schedule == "cosine":
            s = 0.008
            x = torch.linspace(0, n, n + 1)
            ac = torch.cos(((x / n) + s) / (1 + s) * torch.pi * 0.5) ** 2
            ac = ac / ac[0]
            self.betas = torch.clip(1 - (ac[1:] / ac[:-1]), 1e-6, 0.999)
        elif beta_schedule == "sigmoid":
            self.betas = torch.sigmoid(torch.linspace(-6, 6, n)) * (beta_end - beta_start) + beta_start
        else:
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        self.set_alpha_to_one = set_alpha_to_one
        self.num_train_timesteps = n
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.sigmas = (1 / self.alphas_cumprod - 1).sqrt()
        self.final_sigma = (1 / self.final_alpha_cumprod - 1).sqrt()
        self.train_timesteps = torch.tensor(np.arange(0, n).astype(np.int64))
        self.timesteps = self.train_timesteps
        self.sampling_sigmas = self.sigmas
        self.continuous_t = False
        self.sampler_var = sampler_var
        self.eta = eta
        prev = torch.cat([self.final_alpha_cumprod.view(1), self.alphas_cumprod[:-1]])
        self.posterior_variance = self.betas * (1.0 - prev) / (1.0 - self.alphas_cumprod)
        self.min_var_coef = self.posterior_variance[1]
        self.num_inference_steps = n
        self.device = torch.device("cpu")
        self._dev_sigmas = None
        self.reset_state()

    # ---- bookkeeping -------------------------------------------------------------------------
    def to(self, device):
        """The tables stay on the host (they drive the Python loop); a device copy of the sigma table
        feeds the searchsorted kernels."""
        self.device = torch.device(device)
        self._dev_sigmas = None
        self._dev_slopes = None
        return self

    def device_sigmas(self, device=None):
        device = torch.device(device) if device is not None else self.device
        if self._dev_sigmas is None or self._dev_sigmas.device != device:
            self._dev_sigmas = self.sigmas.to(device=device, dtype=torch.float32).contiguous()
        return self._dev_sigmas

    def device_t_slopes(self, device=None):
        """None for discrete schedules; for continuous-t schedules the Interp1d slopes of
        sigma_to_t_interp (src/schedulers.py:210-220; src/torchinterp1d.py: slopes = dy / (eps + dx)), f32."""
        if not self.continuous_t:
            return None
        device = torch.device(device) if device is not None else self.device
        sl = getattr(self, "_dev_slopes", None)
        if sl is None or sl.device != device:
            x, y = self.sigmas.float(), self.train_timesteps.float()
            sl = ((y[1:] - y[:-1]) / (torch.finfo(torch.float32).eps + (x[1:] - x[:-1]))).to(device).contiguous()
            self._dev_slopes = sl
        return sl

    def reset_state(self):
        self.state = {}
        self.i = 0

    # ---- maps (host) ---------------------------------------------------------------------------
    def sigma_to_t(self, sigma):
        sigma = torch.as_tensor(sigma)
        return torch.searchsorted(self.sigmas, sigma.detach().to("cpu", self.sigmas.dtype))

    def t_to_sigma_interp(self, t):
        t = torch.as_tensor(t).cpu()
        xnew = t.squeeze()
        if xnew.dim() == 0:
            xnew = xnew.unsqueeze(0)
        if not xnew.is_floating_point():
            xnew = xnew.float()
        y = _interp1d(self.train_timesteps.float(), self.alphas_cumprod, xnew).squeeze(0)
        return torch.where(t >= 0, (1 / y - 1).sqrt(), self.final_sigma).float()

    def sigma_to_t_interp(self, sigma):
        xnew = torch.as_tensor(sigma).cpu().squeeze()
        if xnew.dim() == 0:
            xnew = xnew.unsqueeze(0)
        return _interp1d(self.sigmas, self.train_timesteps.float(), xnew).squeeze(0).float()   # xnew keeps its dtype (f64 tables promote)

    def sigma(self, timestep):
        timestep = torch.as_tensor(timestep).cpu()
        return torch.where(timestep >= 0, self.sigmas[timestep], self.final_sigma)

    def alpha_bar(self, timestep):
        timestep = torch.as_tensor(timestep).cpu()
        return torch.where(timestep >= 0, self.alphas_cumprod[timestep], self.final_alpha_cumprod)

    def get_sigma(self, timestep):
        return self.t_to_sigma_interp(timestep) if self.continuous_t else self.sigma(timestep)

    def get_alpha_bar(self, timestep):
        if self.continuous_t:
            return 1 / (self.t_to_sigma_interp(timestep) ** 2 + 1)
        return self.alpha_bar(timestep)

    def get_t_from_sigma(self, sigma):
        return self.sigma_to_t_interp(sigma) if self.continuous_t else self.sigma_to_t(sigma)

    def diffusion(self, x_0, t, noise=None):
        """src/schedulers.py:323-329, the training-time forward process  x_n = x_0 sqrt(abar_t) + noise sqrt(1 - abar_t)
        (nlc_lincomb_rows).  ``noise`` defaults to a host draw from the global CPU generator, uploaded (the reference
        draws torch.randn_like on the device; host draws keep seeds reproducible against the CPU reference)."""
        if not x_0.is_cuda:
            raise NlcError("diffusion: tensors must live on the GPU (no CPU fallback)")
        if noise is None:
            noise = torch.randn(x_0.shape, dtype=torch.float32)
        dev = x_0.device
        x_0 = x_0.to(torch.float32).contiguous()
        noise = noise.to(dev, torch.float32).contiguous()
        alpha = self.alphas_cumprod[torch.as_tensor(t).cpu().long().reshape(-1)]             # host table lookup, [B] f32
        if alpha.numel() == 1:
            alpha = alpha.expand(x_0.shape[0])
        with torch.cuda.device(dev):
            x_n = ops.lincomb_rows(x_0, alpha.sqrt().to(dev).contiguous(), noise, (1 - alpha).sqrt().to(dev).contiguous())
        return x_n, noise

    # ---- sampling ladders (src/schedulers.py:227-284) ----------------------------------------------------------------------
    # One builder per sigma style: each returns (timesteps or None, sigmas).  The arithmetic inside them keeps the reference's
    # operation order and dtypes (numpy float64 for the "Linear" / "Scaled" ladders, python floats for "EDM", table lookups for
    # "DDIM") because the resulting tables are compared bit for bit with the reference's (tests/golden/sched.npz).
    def _ladder_ddim(self, start, end, n, scale, dtype):
        lo, hi = self.get_t_from_sigma(end).item(), self.get_t_from_sigma(start).item()
        picked = sorted(space_timesteps(num_timesteps=hi + 1 - lo, section_counts=str(n)), reverse=True)
        t = torch.tensor(lo + np.array(picked), dtype=dtype)
        return t, self.get_sigma(t)

    def _ladder_edm(self, start, end, n, scale, dtype, rho=7):
        a, b = start ** (1 / rho), end ** (1 / rho)
        return None, torch.tensor([(a + i / (n - 1) * (b - a)) ** rho for i in range(n)])

    def _ladder_linear(self, start, end, n, scale, dtype):
        return None, torch.tensor(np.exp(np.linspace(np.log(start), np.log(end), n)))

    def _ladder_scaled(self, start, end, n, scale, dtype):
        steps = np.cumsum(scale ** np.arange(n - 1))                       # geometric step lengths in log sigma
        logs = np.log(start) + (np.log(end) - np.log(start)) / steps[-1] * steps
        return None, torch.tensor(np.exp(np.insert(logs, 0, np.log(start))))

    _LADDERS = {"DDIM": _ladder_ddim, "EDM": _ladder_edm, "Linear": _ladder_linear, "Scaled": _ladder_scaled}

    def set_timesteps_sigma(self, start, end, num_inference_steps, style="DDIM", scale=1, continuous_t=False):
        """The sampling schedule: ``timesteps`` / ``sampling_sigmas`` (+ the terminal (-1, final_sigma) pair when set_alpha_to_one)
        and ``min_var_coef`` from the last real step (src/schedulers.py:227-284)."""
        if style not in self._LADDERS:
            raise ValueError("Invalid style!")
        self.continuous_t = continuous_t
        self.num_inference_steps = num_inference_steps
        n = num_inference_steps + (0 if self.set_alpha_to_one else 1)
        t, sigmas = self._LADDERS[style](self, start, end, n, scale, torch.float32 if continuous_t else torch.long)
        if t is None:
            t = self.get_t_from_sigma(sigmas)
        t, sigmas = t.squeeze(), sigmas.squeeze()
        if not continuous_t:                                               # integer steps: de-duplicate, then sigma is the table's
            t = replace_duplicate_t(t)
            sigmas = self.get_sigma(t)
        if self.set_alpha_to_one:
            t = torch.cat([t, torch.tensor([-1])])
            sigmas = torch.cat([sigmas, torch.tensor([self.final_sigma])])
        self.timesteps, self.sampling_sigmas = t, sigmas
        s_t, s_p = sigmas[-3], sigmas[-2]                                  # the last step that ends at a real noise level
        beta = (s_t ** 2 - s_p ** 2) / (s_t ** 2 + 1)
        self.min_var_coef = beta * (1 - 1 / (s_p ** 2 + 1)) / (1 - 1 / (s_t ** 2 + 1))

    # ---- per-step algebra -------------------------------------------------------------------
    def get_eps_logvar(self, sigma_t, sigma_prev, learned_logvar=None):
        """src/schedulers.py:367-390 on whatever device the inputs live (tiny per-sample tensors; the loop
        itself uses the fused kernel, which recomputes this in registers)."""
        beta_t = ((sigma_t ** 2 - sigma_prev ** 2) / (sigma_t ** 2 + 1)).abs().clamp(min=1e-20)
        coef = ((1 - 1 / (sigma_prev ** 2 + 1)) / (1 - 1 / (sigma_t ** 2 + 1))).clamp(min=0, max=1)
        max_logvar = beta_t.log()
        min_logvar = (beta_t * coef).clamp(min=float(self.min_var_coef)).log()
        if self.sampler_var == "learned":
            frac = (learned_logvar + 1) / 2
            return frac * max_logvar + (1 - frac) * min_logvar
        if self.sampler_var == "fixedsmall":
            return min_logvar
        if self.sampler_var == "fixedlarge":
            return max_logvar
        return None

    @staticmethod
    def _per_sample(v, B, device):
        v = torch.as_tensor(v, dtype=torch.float32)
        v = v.reshape(-1)
        if v.numel() == 1:
            v = v.expand(B)
        return v.to(device).contiguous()

    def pred_xstart(self, xt, eps, sigma_t):
        """xt - sigma_t*eps (src/schedulers.py:407-409) via nlc_sched_x0."""
        if not xt.is_cuda:
            raise NlcError("pred_xstart: tensors must live on the GPU (no CPU fallback)")
        B, Cc = xt.shape[0], xt.shape[1]
        st = self._per_sample(sigma_t, B, xt.device)
        x0 = torch.empty_like(xt, dtype=torch.float32)
        d = SchedDesc(xt=xt.contiguous().data_ptr(), eps_out=eps.contiguous().data_ptr(), sigma_t=st.data_ptr(),
                      sigma_prev=st.data_ptr(), x0=x0.data_ptr(), B=B, C=Cc, Cnet=Cc, HW=xt.numel() // (B * Cc),
                      variant=0, clip=0, var_mode=0, phases=0, eta=0.0, min_var_coef=float(self.min_var_coef))
        with torch.cuda.device(xt.device):
            ops.sched_x0(d)
        return x0

    def pred_xprev(self, x0, eps, sigma_t, sigma_prev, xt=None, log_variance=None, noise=None):
        """The subclass's pred_xprev via nlc_sched_step.  ``noise`` replaces the reference's device-side
        torch.randn_like(x0) (drawn here from the global CPU generator and uploaded, so seeds reproduce)."""
        if self.variant is None:
            raise NotImplementedError
        if not x0.is_cuda:
            raise NlcError("pred_xprev: tensors must live on the GPU (no CPU fallback)")
        B, Cc = x0.shape[0], x0.shape[1]
        dev = x0.device
        st, sp = self._per_sample(sigma_t, B, dev), self._per_sample(sigma_prev, B, dev)
        stochastic = self.eta > 0 or self.variant in ("ddpm", "ddpm_orig")
        if stochastic and noise is None:
            noise = torch.randn(x0.shape, dtype=torch.float32)
        lv = None
        if log_variance is not None:
            lv = torch.as_tensor(log_variance, dtype=torch.float32).to(dev).expand(x0.shape).contiguous()
        x0c = x0.contiguous().clone()
        xp = torch.empty_like(x0c)
        xtc = x0c if xt is None else xt.contiguous()
        d = SchedDesc(xt=xtc.data_ptr(), eps_out=eps.contiguous().data_ptr(),
                      noise=None if noise is None else noise.to(dev).contiguous().data_ptr(),
                      sigma_t=st.data_ptr(), sigma_prev=sp.data_ptr(), logvar_ext=None if lv is None else lv.data_ptr(),
                      x0=x0c.data_ptr(), x_prev=xp.data_ptr(), B=B, C=Cc, Cnet=Cc, HW=x0.numel() // (B * Cc),
                      variant=SCHED_VARIANTS[self.variant], clip=0, var_mode=VAR_MODES[self.sampler_var], phases=2,
                      eta=float(self.eta), min_var_coef=float(self.min_var_coef))
        # keep temporaries alive until the launch is enqueued
        self._keep = (st, sp, lv, noise, x0c, xtc)
        with torch.cuda.device(dev):
            ops.sched_step(d)
        self.i += 1
        return xp


class DDIM_Scheduler(Scheduler):
    variant = "ddim"                 # src/schedulers.py:425-449


class DDIM_simple_Scheduler(Scheduler):
    variant = "ddim_simple"          # :458-473


class DDIM_simple_orig_Scheduler(Scheduler):
    variant = "ddim_simple_orig"     # :480-496


class DDIM_simple_drag_Scheduler(Scheduler):
    variant = "ddim_simple_drag"     # :498-514


class DDPM_Scheduler(Scheduler):
    variant = "ddpm"                 # :541-562

    def __init__(self, *a, eta=1.0, **k):
        super().__init__(*a, eta=eta, **k)


class DDPM_orig_Scheduler(Scheduler):
    variant = "ddpm_orig"            # :573-599 (eta forced to 1)

    def __init__(self, *a, eta=1.0, **k):
        super().__init__(*a, eta=1.0, **k)


class DDIM_orig_Scheduler(Scheduler):
    variant = "ddim_orig"            # :602-627


_SAMPLERS = {"ddpm": DDPM_Scheduler, "ddim": DDIM_Scheduler, "ddim_simple": DDIM_simple_Scheduler,
             "ddim_orig": DDIM_orig_Scheduler, "ddim_simple_orig": DDIM_simple_orig_Scheduler,
             "ddim_simple_drag": DDIM_simple_drag_Scheduler, "ddpm_orig": DDPM_orig_Scheduler}


def get_sampler(sampler_name, train_timesteps, inference_timesteps, beta_start=0.0001, beta_end=0.02,
                beta_schedule="linear", sigma_style="DDIM", set_alpha_to_one=True, start_sigma=None, end_sigma=None,
                sampler_var="none", continuous_t=False, linear_scale=1.0, eta=0.0, ge_gamma=2, norm_eps=False,
                start_t=None, end_t=None):
    """src/schedulers.py:676-726.  'ge' is rejected: the reference's GE_Scheduler.pred_xprev has no ``xt``
    parameter and raises TypeError from denoise_loop."""
    if sampler_name == "ge":
        raise NotImplementedError("sampler 'ge' is broken in the reference (TypeError in denoise_loop) and not provided")
    if sampler_name not in _SAMPLERS:
        raise NotImplementedError
    sampler = _SAMPLERS[sampler_name](num_train_timesteps=train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, set_alpha_to_one=set_alpha_to_one,
                                      sampler_var=sampler_var, eta=eta)
    if start_sigma is None or start_sigma <= 0:
        start_sigma = sampler.sigmas[-1] if (start_t is None or start_t < 0) else min(sampler.sigmas[start_t], sampler.sigmas[-1])
    else:
        start_sigma = torch.tensor(min(start_sigma, sampler.sigmas[-1]))
    if end_sigma is None or end_sigma <= 0:
        end_sigma = sampler.sigmas[0] if (end_t is None or end_t < 0) else sampler.sigmas[end_t]
    sampler.set_timesteps_sigma(start=start_sigma, end=end_sigma, num_inference_steps=inference_timesteps,
                                style=sigma_style, scale=linear_scale, continuous_t=continuous_t)
    return sampler


def redesign_sigma(sampler, num_timesteps, max_T, cycle_size, min_sigma, max_sigma, sigma_gamma):
    """The '--redesign_sigma' tail of the sampling schedule (image_sample.py:788-800): after the ``num_timesteps``
    scheduled steps, ``max_T - num_timesteps`` extra low-noise steps whose log-sigma follows a decaying
    triangle wave between ``min_sigma`` and ``max_sigma`` (period ``cycle_size``, amplitude x ``sigma_gamma``
    per cycle).  Switches the scheduler to continuous t; the resulting table is float64, as in the reference
    (numpy exp -> torch.cat promotes)."""
    if max_T <= num_timesteps:
        return sampler
    sampler.continuous_t = True
    sampler._dev_slopes = None
    it = np.arange(max_T - num_timesteps)
    cycle = np.floor(1 + it / cycle_size)
    x = np.abs(it / cycle_size - cycle + 1)
    lo, hi = np.log(min_sigma), np.log(max_sigma)
    tail = torch.tensor(np.exp(lo + (hi - lo) * np.maximum(0, 1 - x) * sigma_gamma ** (cycle - 1)))
    sampler.sampling_sigmas = torch.cat([torch.clamp(sampler.sampling_sigmas[:-1], min=min_sigma), tail])
    sampler.timesteps = sampler.get_t_from_sigma(sampler.sampling_sigmas)
    sampler.timesteps = torch.cat([sampler.timesteps, torch.tensor([-1])])
    sampler.sampling_sigmas = torch.cat([sampler.sampling_sigmas, torch.tensor([sampler.final_sigma])])
    return sampler
