"""EDM / Heun + NLC sampler on the device (drop-in for EDMImageExperiment, src/experiments.py:756-961).

As in the reference the sampler state and eps are FLOAT64 and the network runs in float32
(:860,872,789-802).  All B x D work is HIP kernels (csrc/edm.hip + the networks); the per-sample sigma
bookkeeping (a handful of [B]-sized f64 values per evaluation) is plain tensor algebra on device
tensors so nothing syncs with the host inside the loop.

    per evaluation   nlc_cast_f64_f32(x)                         x.to(float32)                    (:778,789)
                     [nlc_row_sumsq_f64]                         refine_prior_sigma norm          (:808-815)
                     nlc_edm_scalars                             c_in, c_noise, c_skip, c_out     (:790-797)
                     SongUNet.encode -> SigmaModel (HIP nets)    NLC residual r                   (:822-828)
                     SongUNet.forward (input scaled by c_in in the first conv)                    (:799)
                     nlc_edm_eps                                 denoised, eps = (x-D)/sigma      (:801,836-840)
                     [nlc_row_sumsq_f64 + nlc_f64_lincomb]       normalize(eps)                   (:841-842)
    per step         nlc_f64_lincomb x4-6                        eps rescale, Euler, mix, Heun    (:884-917)
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import numpy as np
import torch

from . import _ext, ops
from ._ext import NlcError, check
from .experiments import ImageExperiment, StackedRandomGenerator, save_image


def _s():
    return torch.cuda.current_stream().cuda_stream


def cast_f64_f32(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(_ext.load().nlc_cast_f64_f32(x.data_ptr(), out.data_ptr(), x.numel(), _s()), "nlc_cast_f64_f32")
    return out


def row_sumsq_f64(x: torch.Tensor) -> torch.Tensor:
    B = x.shape[0]
    out = torch.empty(B, device=x.device, dtype=torch.float64)
    check(_ext.load().nlc_row_sumsq_f64(x.data_ptr(), out.data_ptr(), B, x.numel() // B, _s()), "nlc_row_sumsq_f64")
    return out


def lincomb(x: torch.Tensor, ca: torch.Tensor, y: Optional[torch.Tensor] = None, cb: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ca[b]*x (+ cb[b]*y) in f64; coefficients are [B] f64 device tensors."""
    B = x.shape[0]
    out = torch.empty_like(x)
    ca = ca.to(torch.float64).contiguous()
    if cb is not None:
        cb = cb.to(torch.float64).contiguous()
    check(_ext.load().nlc_f64_lincomb(x.data_ptr(), ca.data_ptr(), None if y is None else y.data_ptr(),
                                      None if cb is None else cb.data_ptr(), out.data_ptr(), B, x.numel() // B, _s()),
          "nlc_f64_lincomb")
    return out


class EDMImageExperiment(ImageExperiment):
    def __init__(self, model, scheduler, batch_size=64, data_shape=(3, 32, 32), seed=0, device="cuda:0", save_folder="./",
                 dist_train=False, time_shift=0, sigma_min=0.002, sigma_max=80, rho=7, S_churn=0, S_min=0,
                 S_max=float("inf"), S_noise=1, sigma_data=0.5, P_mean=-1.2, P_std=1.2, num_timesteps=18):
        super().__init__(model=model, scheduler=scheduler, batch_size=batch_size, data_shape=data_shape, seed=seed,
                         device=device, save_folder=save_folder, dist_train=dist_train, time_shift=time_shift)
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho
        self.S_churn, self.S_min, self.S_max, self.S_noise = S_churn, S_min, S_max, S_noise
        self.sigma_data, self.P_mean, self.P_std = sigma_data, P_mean, P_std
        self.num_timesteps = num_timesteps

    # ---- helpers ----------------------------------------------------------------------------------
    def _per_sample(self, v, B):
        v = torch.as_tensor(v, dtype=torch.float64, device=self.device).reshape(-1)
        return v.expand(B).contiguous() if v.numel() == 1 else v.contiguous()

    def _scalars(self, sigma):
        B = sigma.shape[0]
        f = lambda: torch.empty(B, device=self.device, dtype=torch.float32)
        c_in, c_noise, c_skip, c_out = f(), f(), f(), f()
        check(_ext.load().nlc_edm_scalars(sigma.data_ptr(), float(self.sigma_data), c_in.data_ptr(), c_noise.data_ptr(),
                                          c_skip.data_ptr(), c_out.data_ptr(), B, _s()), "nlc_edm_scalars")
        return c_in, c_noise, c_skip, c_out

    @ops.on_device
    def encode_edm(self, xt, sigma):
        """:777-786 (returns NCHW f32 like the reference)."""
        xt = xt.to(self.device, torch.float64).contiguous()
        c_in, c_noise, _, _ = self._scalars(self._per_sample(sigma, xt.shape[0]))
        return self.model.run(cast_f64_f32(xt), c_noise, mode="encode", in_scale=c_in)

    @ops.on_device
    def pred_edm(self, xt, sigma):
        """:788-802"""
        xt = xt.to(self.device, torch.float64).contiguous()
        B = xt.shape[0]
        sig = self._per_sample(sigma, B)
        c_in, c_noise, c_skip, c_out = self._scalars(sig)
        x32 = cast_f64_f32(xt)
        F = self.model.run(x32, c_noise, mode="forward", in_scale=c_in)
        eps = torch.empty_like(xt)
        den = torch.empty_like(xt)
        check(_ext.load().nlc_edm_eps(xt.data_ptr(), x32.data_ptr(), F.data_ptr(), c_skip.data_ptr(), c_out.data_ptr(),
                                      sig.data_ptr(), eps.data_ptr(), den.data_ptr(), B, xt.numel() // B, _s()), "nlc_edm_eps")
        return den.float()

    @staticmethod
    def _is_scalar(v):
        """The reference's ``len(sigma.unsqueeze(-1)) == 1`` test (src/experiments.py:811,816,829-834): a 0-dim value or a
        one-element tensor."""
        return v.dim() == 0 or v.shape[0] == 1

    def _small(self, v):
        """A sigma-like value as a device tensor with the reference's OWN shape and dtype (0-dim float64 for schedule
        scalars, (B,1,1,1) for per-sample values): the per-step sigma algebra below is then plain torch on a handful of
        elements and inherits ATen's type promotion exactly - a 0-dim f64 operand does NOT promote an f32 (B,1,1,1)
        one, so without refine_prior_sigma the NLC products of the reference are float32 (src/experiments.py:824-825)."""
        if torch.is_tensor(v):
            return v.to(self.device)
        return torch.as_tensor(v, dtype=torch.float64, device=self.device)

    @torch.no_grad()
    @ops.on_device
    def get_denoise_vector(self, xt, sigma_t, sigma_prev, style="base", norm_eps=False, refine_prior_sigma=False):
        """:804-843, op by op.  xt: f64 [B,C,H,W] on the device; sigma_t / sigma_prev: scalars (0-dim) or (B,1,1,1).
        Returns (eps f64, denoised f64, sigma_t, sigma_prev) with the reference's shapes ((1,1,1,1) for scalars) and dtypes."""
        B = xt.shape[0]
        D = xt.numel() // B
        sigma_t, sigma_prev = self._small(sigma_t), self._small(sigma_prev)
        sigma_t_orig = sigma_t
        if refine_prior_sigma:
            norm_x = (row_sumsq_f64(xt).sqrt() / math.sqrt(self.dim)).view(B, 1, 1, 1)
            min_dist = torch.clamp(norm_x - self.norm_max, min=0)
            max_dist = norm_x + self.norm_min
            raw_sigma = torch.ones_like(norm_x) * sigma_t if self._is_scalar(sigma_t) else sigma_t
            sigma_t = torch.minimum(torch.maximum(raw_sigma, min_dist), max_dist)      # torch.clamp(min=, max=) order
            if self._is_scalar(sigma_prev):
                sigma_prev = torch.ones_like(norm_x) * sigma_prev
        x32 = cast_f64_f32(xt)
        if "pred" in style:
            if self.sigma_model is None:
                raise NlcError("style '%s' needs a sigma model (set_model)" % style)
            c_in, c_noise, _, _ = self._scalars(self._per_sample(sigma_t, B))
            feat = self.model.run(x32, c_noise, mode="encode", in_scale=c_in, feat_nhwc=True)
            sigma_residual = self.sigma_model.run_nhwc(feat).view(B, 1, 1, 1)          # f32
            dist_hat = sigma_t * (1 + sigma_residual)
            dist_prev_hat = dist_hat * (sigma_prev / sigma_t)
            sigma_t = dist_hat
            if style == "pred":
                sigma_prev = dist_prev_hat
        if self._is_scalar(sigma_t_orig):
            sigma_t_orig = sigma_t_orig.reshape(-1, 1, 1, 1)
        if self._is_scalar(sigma_t):
            sigma_t = sigma_t.reshape(-1, 1, 1, 1)
        if self._is_scalar(sigma_prev):
            sigma_prev = sigma_prev.reshape(-1, 1, 1, 1)
        sig_div = self._per_sample(sigma_t_orig if style == "pred_sigma" else sigma_t, B)
        c_in, c_noise, c_skip, c_out = self._scalars(sig_div)
        F = self.model.run(x32, c_noise, mode="forward", in_scale=c_in)
        eps, den = torch.empty_like(xt), torch.empty_like(xt)
        check(_ext.load().nlc_edm_eps(xt.data_ptr(), x32.data_ptr(), F.data_ptr(), c_skip.data_ptr(), c_out.data_ptr(),
                                      sig_div.data_ptr(), eps.data_ptr(), den.data_ptr(), B, D, _s()), "nlc_edm_eps")
        if norm_eps:
            eps = self._normalize(eps)
        return eps, den, sigma_t, sigma_prev

    def _normalize(self, x):
        """utils.normalize on the f64 tensor (:841-842,908-909)."""
        denom = torch.clamp(row_sumsq_f64(x).sqrt(), min=1e-12)
        return lincomb(x, math.sqrt(self.dim) / denom)

    @torch.no_grad()
    @ops.on_device
    def edm_sampler(self, shape, gen=None, style="base,base", norm_eps="000", refine_prior_sigma=False, num_steps=None,
                    sigma_scheduler="EDM", eps_ratio=0.5, eps_scale=1.0, use_second_order=True, latents=None):
        """:846-918.  ``latents`` (optional, host or device N(0,1)) replaces ``gen.randn(shape)``."""
        norm_e, norm_combine = bool(int(norm_eps[0])), bool(int(norm_eps[1]))
        style_t, style_next = style.split(",")
        n = self.num_timesteps if num_steps is None else num_steps
        if latents is None:
            latents = gen.randn(shape, device=self.device)
        latents = latents.to(self.device)
        B = latents.shape[0]
        idx = torch.arange(n, dtype=torch.float64)
        if sigma_scheduler == "EDM":
            steps = (self.sigma_max ** (1 / self.rho) + idx / (n - 1) * (self.sigma_min ** (1 / self.rho) - self.sigma_max ** (1 / self.rho))) ** self.rho
        elif sigma_scheduler == "Linear":
            steps = torch.tensor(np.exp(np.linspace(np.log(self.sigma_max), np.log(self.sigma_min), n)))
        else:
            raise NotImplementedError
        steps = torch.cat([torch.as_tensor(steps), torch.zeros_like(steps[:1])])          # host f64 schedule
        steps_dev = steps.to(self.device)                                              # 0-dim views of it feed the sigma algebra
        one = torch.ones(B, device=self.device, dtype=torch.float64)
        co = lambda v: self._per_sample(v, B)                                          # sigma-like -> [B] f64 kernel coefficients
        x_next = lincomb(latents.to(torch.float64).contiguous(), one * float(steps[0]))
        for i, (s_cur, s_next) in enumerate(zip(steps[:-1], steps[1:])):
            x_cur = x_next
            s_cur = float(s_cur)
            sigma_next0 = steps_dev[i + 1]                                             # 0-dim f64, as in the reference
            gamma = min(self.S_churn / n, np.sqrt(2) - 1) if self.S_min <= s_cur <= self.S_max else 0
            sigma_hat0 = steps_dev[i] + gamma * steps_dev[i]
            # churn noise (:880): the reference draws randn_like on EVERY step from the global generator; with
            # S_churn > 0 the draw is kept on every step so the stream stays aligned when S_min/S_max gate gamma
            z = torch.randn(x_cur.shape, dtype=torch.float64) if self.S_churn > 0 else None
            if gamma > 0:
                z = z.to(self.device)
                x_hat = lincomb(x_cur, one, z, co((sigma_hat0 ** 2 - steps_dev[i] ** 2).sqrt() * self.S_noise))
            else:
                x_hat = x_cur                                                  # + 0 * randn_like: exact no-op
            eps, _, sigma_hat, sigma_next = self.get_denoise_vector(x_hat, sigma_hat0, sigma_next0, style=style_t, norm_eps=norm_e,
                                                                    refine_prior_sigma=refine_prior_sigma)
            eps = lincomb(eps, co(sigma_hat / sigma_hat0))                     # eps * (sigma_hat / sigma_hat0)  (:884)
            if "pred_partial" in style_t:
                sigma_next = sigma_next0
            if style_t == "pred_partial":
                x_next = lincomb(x_hat, one, eps, co(sigma_next - sigma_hat0))
            else:
                x_next = lincomb(x_hat, one, eps, co(sigma_next - sigma_hat))
            if style_t == "pred_partial3":
                sigma_hat = sigma_hat0
            if i < n - 1 and use_second_order:
                eps_next, _, sigma_next, _ = self.get_denoise_vector(x_next, sigma_next, sigma_next * 0, style=style_next,
                                                                     norm_eps=norm_e, refine_prior_sigma=refine_prior_sigma)
                eps_next = lincomb(eps_next, co(sigma_next / sigma_next0))     # (:904)
                if "pred_partial" in style_next:
                    sigma_next = sigma_next0
                new_eps = lincomb(eps, one * eps_ratio, eps_next, one * (1 - eps_ratio))
                if norm_combine:
                    new_eps = self._normalize(new_eps)
                if eps_scale is not None:
                    new_eps = lincomb(new_eps, one / eps_scale)
                else:                                                          # cosine-similarity scaling (:911-916)
                    cs = torch.empty(B, device=self.device, dtype=torch.float64)
                    check(_ext.load().nlc_row_cosine_f64(new_eps.data_ptr(), eps.data_ptr(), 1e-6, cs.data_ptr(), B,
                                                         new_eps.numel() // B, _s()), "nlc_row_cosine_f64")
                    new_eps = lincomb(new_eps, cs)
                x_next = lincomb(x_hat, one, new_eps, co(sigma_next - sigma_hat))
        return x_next

    @torch.no_grad()
    def evaluate_edm(self, n_samples, images_dir, gen=None, style="base,base", norm_eps="000", refine_prior_sigma=False,
                     microbatch=-1, sigma_scheduler="EDM", eps_ratio=0.5, eps_scale=1.0, use_second_order=True,
                     return_samples=False, save_images=True):
        """:922-961, same positional arguments and the same return value: ``log_dict`` (``{'fid': ...}``).

        Per-batch host generators seeded with the global sample indices (StackedRandomGenerator), samples mapped to [0,1] and
        written as ``{rank:02}-{batch:05}-{j:03}.png`` into ``images_dir``; a batch whose PNGs all exist is skipped (:935-943);
        FID over ``images_dir`` (NaN without pytorch_fid).  ``gen`` is accepted and ignored, as upstream (:933 overwrites it).
        Extensions (keywords only): ``return_samples=True`` returns ``(log_dict, samples)``; the [0,1] samples of the batches
        this call sampled are also left in ``self.last_samples`` (float64, in batch order); ``save_images=False`` writes no PNGs
        (``images_dir`` may then be None).

        Under a launcher (one process per GPU, shard.init_from_env) rank r samples the to-do batches r, r+W, ...: every sample
        owns its generator (seed = global sample index), so nothing has to be replayed; one all-gather collects the samples in
        order on every rank, and rank 0 writes the PNGs (prefix 00, as the single-process run) and computes FID."""
        from . import shard
        world, rank = shard.world_rank()
        batch_size = microbatch if microbatch > 0 else self.batch_size
        if n_samples % batch_size:
            raise ValueError("n_samples must be a multiple of batch_size (the reference asserts this at :77, SURVEY.md §9)")
        n_batches = n_samples // batch_size
        seeds = torch.arange(n_samples).tensor_split(n_batches)
        write = bool(save_images and images_dir)
        paths = [[os.path.join(images_dir, f"{0:02}-{i:05}-{j:03}.png") for j in range(batch_size)] if write else []
                 for i in range(n_batches)]
        skip = shard.broadcast_object([bool(write and all(os.path.exists(p) for p in pp)) for pp in paths])
        for i in range(n_batches):
            if skip[i]:
                print("skip images for:", f"{0:02}-{i:05}-({0:03}~{batch_size - 1:03}).png")
        todo = [i for i in range(n_batches) if not skip[i]]
        outs = []
        for k, i in enumerate(todo):
            if k % world != rank:
                continue
            g = StackedRandomGenerator(self.device, seeds[i])
            x = self.edm_sampler(shape=(batch_size,) + self.data_shape, gen=g, style=style, norm_eps=norm_eps,
                                 refine_prior_sigma=refine_prior_sigma, sigma_scheduler=sigma_scheduler, eps_ratio=eps_ratio,
                                 eps_scale=eps_scale, use_second_order=use_second_order)
            sample = x.add(1).div(2).clamp(0, 1)
            if write and world == 1:
                for img, p in zip(sample, paths[i]):
                    save_image(img, p)
            outs.append(sample)
            print(f"done batches:{i}/{n_batches}")
        if world > 1:
            local = torch.stack(outs) if outs else torch.empty((0, batch_size) + tuple(self.data_shape), device=self.device, dtype=torch.float64)
            allx = shard.gather_samples(local, len(todo), world, rank)
            if write and rank == 0:
                for k, i in enumerate(todo):
                    for img, p in zip(allx[k], paths[i]):
                        save_image(img, p)
            shard.barrier()
            outs = list(allx)
        self.last_samples = torch.cat(outs) if outs else torch.empty((0,) + tuple(self.data_shape), device=self.device, dtype=torch.float64)
        fid = self.fid_fn(images_dir) if (self.fid_fn is not None and images_dir and rank == 0) else (float("nan") if rank == 0 else 0.0)
        log_dict = {"fid": fid}
        return (log_dict, self.last_samples) if return_samples else log_dict
