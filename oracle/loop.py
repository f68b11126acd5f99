"""Oracle: the DDIM+NLC sampling loop and the EDM/Heun+NLC sampler (src/experiments.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Networks come in as plain callables
``eps_fn(x, t)``, ``encode_fn(x, t)``, ``sigma_fn(feat)`` (oracle nets or anything else).
"""
from __future__ import annotations

import math
from itertools import pairwise
from typing import Callable, Optional

import numpy as np
import torch

from .sched import Schedule


def vector_norm(x, keepdim=True):
    """src/utils.py:7-9"""
    return torch.linalg.vector_norm(x, dim=tuple(range(1, x.dim())), keepdim=keepdim)


def normalize(x, inp_dim, eps=1e-12):
    """src/utils.py:11-16: sqrt(D) * x / max(||x||, eps)"""
    return math.sqrt(inp_dim) * x / torch.clamp(vector_norm(x, keepdim=True), min=eps)


def make_clip_fn(kind: str):
    """set_clip_fn (src/experiments.py:186-207)."""
    if kind == "clamp":
        return lambda x: x.clamp(-1, 1)
    if kind == "dynamic":
        def thr(sample, ratio=0.99, max_value=100):
            b = sample.shape[0]
            flat = sample.reshape(b, -1)
            s = torch.quantile(flat.abs(), ratio, dim=1)
            s = torch.clamp(s, min=1, max=max_value).unsqueeze(1)
            return (torch.clamp(flat, -s, s) / s).reshape(sample.shape)
        return thr
    return lambda x: x


class DiffusionOracle:
    """ExperimentDiffusion's sampling half (src/experiments.py:87-102,176-207,255-460)."""

    def __init__(self, eps_fn: Callable, encode_fn: Optional[Callable], sigma_fn: Optional[Callable], sched: Schedule,
                 data_shape, learn_epsvar: bool, norm_min=None, norm_max=None, clip_fn="none", time_shift=0):
        self.eps_fn, self.encode_fn, self.sigma_fn, self.s = eps_fn, encode_fn, sigma_fn, sched
        self.data_shape = tuple(data_shape)
        self.dim = int(np.prod(data_shape))
        self.learn_epsvar = learn_epsvar
        self.time_shift = time_shift
        # set_norm_maxmin (:176-184)
        self.norm_min = norm_min / math.sqrt(self.dim) if norm_min is not None else 0.0
        self.norm_max = norm_max / math.sqrt(self.dim) if norm_max is not None else 1.0
        self.clip = make_clip_fn(clip_fn)

    @staticmethod
    def to_z(xt, sigma):
        """convert_coordinate (:273-282)"""
        return xt * (1 / (sigma ** 2 + 1)).sqrt()

    def _batched_t(self, t, n):
        return torch.ones(n, dtype=int) * t                    # batched_t (:255-258)

    def _net(self, fn, xt, t, sigma, batch_t):
        """pred_xt / encode_xt (:295-311)"""
        zt = self.to_z(xt, sigma)
        if batch_t:
            t = self._batched_t(t, len(xt))
        while t.dim() > 1:
            t = torch.squeeze(t, dim=-1)
        return fn(zt, t)

    @torch.no_grad()
    def get_denoise_vector(self, xt, t, sigma_t, sigma_prev, style="base", norm_eps=False, refine_prior_sigma=False):
        """src/experiments.py:399-460 (chunk_size=1 as the entry points pass it)."""
        if refine_prior_sigma:
            norm_x = vector_norm(xt, keepdim=True) / math.sqrt(self.dim)
            min_dist = torch.clamp(norm_x - self.norm_max, min=0)
            max_dist = norm_x + self.norm_min
            raw_sigma = torch.ones_like(norm_x) * sigma_t if sigma_t.dim() == 0 or len(sigma_t.unsqueeze(-1)) == 1 else sigma_t
            sigma_t = torch.clamp(raw_sigma, min=min_dist, max=max_dist)
            t = self.s.get_t_from_sigma(sigma_t)
            if t.min() > 0:
                t = t - self.time_shift
            if len(sigma_prev.unsqueeze(-1)) == 1:
                sigma_prev = torch.ones_like(norm_x) * sigma_prev
        t = torch.clamp(t, min=0.0, max=1000.0)
        if "pred" in style:
            is_short = t.dim() <= 1
            feat = self._net(self.encode_fn, xt, t, sigma_t, batch_t=is_short)
            r = self.sigma_fn(feat)
            dist_hat = sigma_t * (1 + r)
            dist_prev_hat = dist_hat * (sigma_prev / sigma_t)
            t = torch.clamp(self.s.get_t_from_sigma(dist_hat), min=0.0, max=1000.0)
            sigma_t = dist_hat
            if style == "pred":
                sigma_prev = dist_prev_hat
        elif len(t.unsqueeze(-1)) == 1:
            t = self._batched_t(t, len(xt))
        eps_out = self._net(self.eps_fn, xt, t, sigma_t, batch_t=False)
        if self.learn_epsvar:
            c = eps_out.size(1) // 2
            eps_mean, eps_logvar = torch.split(eps_out, c, dim=1)
        else:
            eps_mean, eps_logvar = eps_out, None
        if norm_eps:
            eps_mean = normalize(eps_mean, self.dim)
        eps_logvar = self.s.get_eps_logvar(sigma_t=sigma_t, sigma_prev=sigma_prev, learned_logvar=eps_logvar)
        return eps_mean, eps_logvar, sigma_t, sigma_prev

    @torch.no_grad()
    def denoise_loop(self, shape, gen=None, style="base", constrain_fn=None, norm_eps=False, refine_prior_sigma=False,
                     xT=None, sigma_pred_threshold=1000, new_eta=None, trace=None, noise_list=None):
        """src/experiments.py:329-397 without the logging side channel; returns the last (clipped) x0.

        trace: optional dict that receives per-step sigma_t / sigma_prev / x0 / xt lists.
        noise_list: optional per-step noise tensors (host-ordered injection for eta > 0)."""
        S = self.s
        if xT is None:
            z = torch.randn(shape, generator=gen)                                  # get_noise (:263-271)
            sigma0 = S.sampling_sigmas[0]
            xt = z / (1 / (sigma0 ** 2 + 1)).sqrt()                                 # inv_convert_coordinate (:284-293)
        else:
            xt = xT
        eta0 = S.eta
        x0 = xt
        for ind, (t, t_prev) in enumerate(pairwise(S.timesteps)):
            if ind == S.num_inference_steps - 1 and new_eta is not None:
                S.eta = new_eta
            sigma_t, sigma_prev = S.sampling_sigmas[ind], S.sampling_sigmas[ind + 1]
            cur_style, cur_refine = style, refine_prior_sigma
            if t > sigma_pred_threshold:
                cur_style, cur_refine = "base", False
            eps, logvar, sigma_t, sigma_prev = self.get_denoise_vector(xt, t, sigma_t, sigma_prev, cur_style, norm_eps, cur_refine)
            x0_hat = self.clip(S.pred_xstart(xt, eps, sigma_t))
            x0 = constrain_fn(x0_hat) if constrain_fn is not None else x0_hat
            noise = None if noise_list is None else noise_list[ind]
            xt = S.pred_xprev(x0=x0, eps=eps, sigma_t=sigma_t, sigma_prev=sigma_prev, xt=xt, log_variance=logvar, noise=noise)
            if trace is not None:
                trace.setdefault("sigma_t", []).append(torch.as_tensor(sigma_t).reshape(-1).clone())
                trace.setdefault("sigma_prev", []).append(torch.as_tensor(sigma_prev).reshape(-1).clone())
                trace.setdefault("x0", []).append(x0.clone())
                trace.setdefault("xt", []).append(xt.clone())
            if torch.isnan(xt).any():
                break
        S.eta = eta0
        return x0


class EdmOracle:
    """EDMImageExperiment's sampling half (src/experiments.py:756-918)."""

    def __init__(self, eps_fn, encode_fn, sigma_fn, data_shape, sigma_min=0.002, sigma_max=80, rho=7, S_churn=0,
                 S_min=0, S_max=float("inf"), S_noise=1, sigma_data=0.5, num_timesteps=18, norm_min=None, norm_max=None):
        self.eps_fn, self.encode_fn, self.sigma_fn = eps_fn, encode_fn, sigma_fn
        self.dim = int(np.prod(data_shape))
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho
        self.S_churn, self.S_min, self.S_max, self.S_noise = S_churn, S_min, S_max, S_noise
        self.sigma_data, self.num_timesteps = sigma_data, num_timesteps
        self.norm_min = norm_min / math.sqrt(self.dim) if norm_min is not None else 0.0
        self.norm_max = norm_max / math.sqrt(self.dim) if norm_max is not None else 1.0

    def encode_edm(self, xt, sigma):
        """:777-786"""
        xt = xt.to(torch.float32)
        sigma = sigma.to(torch.float32).reshape(-1, 1, 1, 1)
        c_in = 1 / (self.sigma_data ** 2 + sigma ** 2).sqrt()
        return self.encode_fn((c_in * xt).to(torch.float32), (sigma.log() / 4).flatten())

    def pred_edm(self, xt, sigma):
        """:788-802"""
        xt = xt.to(torch.float32)
        sigma = sigma.to(torch.float32).reshape(-1, 1, 1, 1)
        sd = self.sigma_data
        c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
        c_out = sigma * sd / (sigma ** 2 + sd ** 2).sqrt()
        c_in = 1 / (sd ** 2 + sigma ** 2).sqrt()
        F_x = self.eps_fn((c_in * xt).to(torch.float32), (sigma.log() / 4).flatten())
        return c_skip * xt + c_out * F_x.to(torch.float32)

    @torch.no_grad()
    def get_denoise_vector(self, xt, sigma_t, sigma_prev, style="base", norm_eps=False, refine_prior_sigma=False):
        """:804-843"""
        sigma_t_orig = sigma_t
        if refine_prior_sigma:
            norm_x = vector_norm(xt, keepdim=True) / math.sqrt(self.dim)
            min_dist = torch.clamp(norm_x - self.norm_max, min=0)
            max_dist = norm_x + self.norm_min
            raw = torch.ones_like(norm_x) * sigma_t if len(sigma_t.unsqueeze(-1)) == 1 else sigma_t
            sigma_t = torch.clamp(raw, min=min_dist, max=max_dist)
            if len(sigma_prev.unsqueeze(-1)) == 1:
                sigma_prev = torch.ones_like(norm_x) * sigma_prev
        if "pred" in style:
            r = self.sigma_fn(self.encode_edm(xt, sigma=sigma_t))
            dist_hat = sigma_t * (1 + r)
            dist_prev_hat = dist_hat * (sigma_prev / sigma_t)
            sigma_t = dist_hat
            if style == "pred":
                sigma_prev = dist_prev_hat
        if len(sigma_t_orig.unsqueeze(-1)) == 1:
            sigma_t_orig = sigma_t_orig.reshape(-1, 1, 1, 1)
        if len(sigma_t.unsqueeze(-1)) == 1:
            sigma_t = sigma_t.reshape(-1, 1, 1, 1)
        if len(sigma_prev.unsqueeze(-1)) == 1:
            sigma_prev = sigma_prev.reshape(-1, 1, 1, 1)
        if style == "pred_sigma":
            denoised = self.pred_edm(xt, sigma_t_orig).to(torch.float64)
            eps = (xt - denoised) / sigma_t_orig
        else:
            denoised = self.pred_edm(xt, sigma_t).to(torch.float64)
            eps = (xt - denoised) / sigma_t
        if norm_eps:
            eps = normalize(eps, self.dim)
        return eps, denoised, sigma_t, sigma_prev

    @torch.no_grad()
    def edm_sampler(self, latents, style="base,base", norm_eps="000", refine_prior_sigma=False, num_steps=None,
                    eps_ratio=0.5, eps_scale=1.0, use_second_order=True, churn_noise=None):
        """:846-918 with sigma_scheduler='EDM'.  ``latents`` is the host-drawn N(0,1) start (the reference draws it
        from per-sample generators, :856); churn_noise optionally replaces torch.randn_like (coefficient is 0 when
        S_churn=0 but the draw still happens, :880)."""
        norm_e, norm_combine = bool(int(norm_eps[0])), bool(int(norm_eps[1]))
        style_t, style_next = style.split(",")
        n = self.num_timesteps if num_steps is None else num_steps
        idx = torch.arange(n, dtype=torch.float64)
        steps = (self.sigma_max ** (1 / self.rho) + idx / (n - 1) * (self.sigma_min ** (1 / self.rho) - self.sigma_max ** (1 / self.rho))) ** self.rho
        steps = torch.cat([torch.as_tensor(steps), torch.zeros_like(steps[:1])])
        sim = torch.nn.CosineSimilarity(dim=1, eps=1e-6)
        x_next = latents.to(torch.float64) * steps[0]
        for i, (s_cur, s_next) in enumerate(zip(steps[:-1], steps[1:])):
            x_cur = x_next
            s_next0 = s_next
            gamma = min(self.S_churn / n, np.sqrt(2) - 1) if self.S_min <= s_cur <= self.S_max else 0
            s_hat = torch.as_tensor(s_cur + gamma * s_cur)
            s_hat0 = s_hat
            z = torch.randn_like(x_cur) if churn_noise is None else churn_noise[i]
            x_hat = x_cur + (s_hat ** 2 - s_cur ** 2).sqrt() * self.S_noise * z
            eps, _, s_hat, s_next = self.get_denoise_vector(x_hat, s_hat, s_next, style=style_t, norm_eps=norm_e,
                                                            refine_prior_sigma=refine_prior_sigma)
            eps = eps * (s_hat / s_hat0)
            if "pred_partial" in style_t:
                s_next = s_next0
            if style_t == "pred_partial":
                x_next = x_hat + (s_next - s_hat0) * eps
            else:
                x_next = x_hat + (s_next - s_hat) * eps
            if style_t == "pred_partial3":
                s_hat = s_hat0
            if i < n - 1 and use_second_order:
                eps_next, _, s_next, _ = self.get_denoise_vector(x_next, s_next, s_next * 0, style=style_next,
                                                                 norm_eps=norm_e, refine_prior_sigma=refine_prior_sigma)
                eps_next = eps_next * (s_next / s_next0)
                if "pred_partial" in style_next:
                    s_next = s_next0
                new_eps = eps_ratio * eps + (1 - eps_ratio) * eps_next
                if norm_combine:
                    new_eps = normalize(new_eps, self.dim)
                if eps_scale is not None:
                    new_eps = new_eps / eps_scale
                else:
                    b = len(new_eps)
                    new_eps = new_eps * sim(new_eps.reshape(b, -1), eps.reshape(b, -1)).reshape(b, 1, 1, 1)
                x_next = x_hat + (s_next - s_hat) * new_eps
        return x_next
