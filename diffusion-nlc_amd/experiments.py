"""Sampling loops on the GPU (drop-in for the sampling half of src/experiments.py).

``ImageExperiment.denoise_loop`` keeps the reference's signature and return value
(src/experiments.py:329-397) but runs every step on the device without host round trips:

    per step   nlc_row_sumsq(xt)                       ||xt||^2 per sample           (refine_prior_sigma)
               nlc_refine_sigma                        sigma clamp, t lookup, c_in    (:401-419)
               UNet.encode  -> SigmaModel  (HIP nets)  NLC residual r                 (:420-424)
               nlc_sigma_correct                       sigma_hat, sigma_prev_hat, t   (:425-431)
               UNet.forward (HIP net, input scaled by c_in inside the first conv)     (:436-450)
               nlc_row_sumsq(eps)                      ||eps||^2 per sample           (norm_eps, :457-458)
               nlc_sched_x0 [+ nlc_dynamic_threshold]  x0_hat (and its 0.99-quantile) (:360-361)
               nlc_sched_step                          clip, x_prev                   (:361-370)

The per-sample scalars never leave the device; the only host syncs are the ones the reference also
has: the optional per-step logging copies and the NaN early-break flag (one int per step).
Noise is drawn on the host from the caller's generator in the reference's order and uploaded.
"""
from __future__ import annotations

import math
from itertools import pairwise
from typing import Callable, Optional

import numpy as np
import torch

from . import ops
from ._ext import CLIP_MODES, SCHED_VARIANTS, VAR_MODES, NlcError, SchedDesc


class StackedRandomGenerator:
    """src/experiments.py:71-85, with the per-sample generators on the HOST so that results are
    reproducible against the CPU reference (SURVEY.md §7 'RNG parity'); draws are uploaded."""

    def __init__(self, device, seeds):
        self.device = torch.device(device)
        self.generators = [torch.Generator().manual_seed(int(seed) % (1 << 32)) for seed in seeds]

    def randn(self, size, **kwargs):
        assert size[0] == len(self.generators)
        kwargs.pop("device", None)
        return torch.stack([torch.randn(size[1:], generator=gen, **kwargs) for gen in self.generators]).to(self.device)

    def randn_like(self, input):
        return self.randn(input.shape, dtype=input.dtype)


def save_image(img, path):
    """torchvision.utils.save_image for one [C,H,W] image in [0,1] (what the reference calls per sample, src/experiments.py:40,950;
    image_sample.py:15): round-to-nearest 8-bit PNG through PIL.  A side effect outside the sampling path; without PIL nothing
    is written."""
    try:
        from PIL import Image
    except ImportError:
        return
    arr = img.detach().to("cpu", torch.float32).mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr[..., 0] if arr.shape[-1] == 1 else arr).save(path)


class ExperimentDiffusion:
    def __init__(self, model, scheduler, batch_size, data_shape, save_folder, seed=0, device="cpu", dist_train=0,
                 time_shift=0):
        self.model = model
        self.scheduler = scheduler
        self.device = torch.device(device)
        self.seed = seed
        self.batch_size = batch_size
        self.data_shape = tuple(data_shape)
        self.shape = (batch_size,) + self.data_shape
        self.dim = int(np.prod(data_shape))
        self.dim_coord = len(data_shape)
        self.save_folder = save_folder
        self.dist_train = dist_train
        self.time_shift = time_shift
        self.clip_kind = "none"
        self.clip_denoise_fn = lambda x: x
        self.learn_epsvar = False
        self.sigma_model = None
        self.norm_min, self.norm_max = 0.0, 1.0
        self.fid_fn = None
        self.gen = self.new_gen()

    # ---- hipGraph replay of the network evaluations (HipModule._graphable) ------------------------
    @property
    def use_graphs(self):
        return bool(getattr(self.model, "use_graphs", False))

    @use_graphs.setter
    def use_graphs(self, flag):
        for m in (self.model, self.sigma_model):
            if m is not None and hasattr(m, "drop_graphs"):
                m.use_graphs = bool(flag)

    # ---- configuration (src/experiments.py:104-114,176-226) -------------------------------------
    def set_model(self, model=None, sigma_model=None, learn_epsvar=True):
        if model is not None:
            self.model = model
            self.learn_epsvar = learn_epsvar
        else:
            self.learn_epsvar = False
        if sigma_model is not None:
            self.sigma_model = sigma_model

    def set_norm_maxmin(self, norm_min=None, norm_max=None):
        self.norm_min = norm_min / math.sqrt(self.dim) if norm_min is not None else 0.0
        self.norm_max = norm_max / math.sqrt(self.dim) if norm_max is not None else 1.0

    def set_clip_fn(self, clip_fn="none"):
        """'clamp' | 'dynamic' (0.99-quantile thresholding, max 100) | anything else = identity."""
        self.clip_kind = clip_fn if clip_fn in ("clamp", "dynamic") else "none"
        self.clip_denoise_fn = self._clip_tensor

    @ops.on_device
    def _clip_tensor(self, x):
        if self.clip_kind == "clamp":
            return x.clamp(-1, 1)
        if self.clip_kind == "dynamic":
            s = ops.dynamic_threshold(x.contiguous(), 0.99, 100.0).view(-1, *([1] * (x.dim() - 1)))
            return torch.clamp(x, -s, s) / s
        return x

    def fid_helper(self, fid_target, dims=2048):
        """FID needs pytorch_fid + an InceptionV3 download; optional here: without the
        package or a target file the metric is reported as NaN instead of failing the run."""
        try:
            from pytorch_fid.fid_score import calculate_frechet_distance, compute_statistics_of_path
            from pytorch_fid.inception import InceptionV3
            with np.load(fid_target) as f:
                m1, s1 = f["mu"][:], f["sigma"][:]
            model = InceptionV3([InceptionV3.BLOCK_INDEX_BY_DIM[dims]]).to(self.device)
            self.fid_fn = lambda path: calculate_frechet_distance(m1, s1, *compute_statistics_of_path(path, model, 128, dims, self.device, 1))
        except Exception:                         # noqa: BLE001 - optional dependency
            self.fid_fn = lambda path: float("nan")

    # ---- small helpers (src/experiments.py:255-325) ----------------------------------------------
    def batched_t(self, t, batch_size=None):
        return torch.ones(self.batch_size if batch_size is None else batch_size, dtype=int, device=self.device) * t

    def new_gen(self, seed=None):
        return torch.manual_seed(self.seed if seed is None else seed)

    def host_draws_per_batch(self, new_eta=None) -> int:
        """How many ``randn(batch_shape)`` one ``denoise_loop`` batch takes from the host generator: the initial state plus, for
        stochastic samplers, one per timestep (``new_eta`` switches the last step, src/experiments.py:349-350).  A sharded run
        replays this many draws for every batch another rank owns (shard.replay_draws) so that the single generator of
        image_sample.py:529 stays in step with the single-process run."""
        S = self.scheduler
        n = 1
        steps = len(S.timesteps) - 1
        for ind in range(steps):
            eta = new_eta if (ind == S.num_inference_steps - 1 and new_eta is not None) else S.eta
            if eta > 0 or S.variant in ("ddpm", "ddpm_orig"):
                n += 1
        return n

    @ops.on_device
    def get_noise(self, shape=None, gen=None, norm_noise=False):
        noise = torch.randn(self.shape if shape is None else shape, generator=self.gen if gen is None else gen)
        noise = noise.to(self.device)
        if norm_noise:
            ss = ops.row_sumsq(noise)
            noise = ops.scale_rows(noise, math.sqrt(self.dim) / torch.clamp(ss.sqrt(), min=1e-12))
        return noise

    @ops.on_device
    def convert_coordinate(self, xt, sigma=None, t=None):
        alpha_bar = 1 / (torch.as_tensor(sigma) ** 2 + 1) if sigma is not None else self.scheduler.get_alpha_bar(t)
        return self._scale(xt, alpha_bar.sqrt())

    @ops.on_device
    def inv_convert_coordinate(self, zt, sigma=None, t=None):
        alpha_bar = 1 / (torch.as_tensor(sigma) ** 2 + 1) if sigma is not None else self.scheduler.get_alpha_bar(t)
        return self._scale(zt, 1 / alpha_bar.sqrt())

    def _scale(self, x, s):
        s = torch.as_tensor(s, dtype=torch.float32).reshape(-1)
        if s.numel() == 1:
            return ops.scale_rows(x.contiguous(), None, float(s))
        return ops.scale_rows(x.contiguous(), s.to(x.device).contiguous(), 1.0)

    @ops.on_device
    def get_noise_xt(self, shape=None, gen=None, norm_noise=False, t=None, sigma=None):
        """Initial state: drawn AND scaled on the host exactly as the reference does (z / sqrt(alpha_bar),
        src/experiments.py:284-293,322-325), then uploaded once per batch."""
        z = torch.randn(self.shape if shape is None else shape, generator=self.gen if gen is None else gen)
        if norm_noise:
            z = math.sqrt(self.dim) * z / torch.clamp(torch.linalg.vector_norm(z, dim=tuple(range(1, z.dim())), keepdim=True), min=1e-12)
        alpha_bar = 1 / (torch.as_tensor(sigma) ** 2 + 1) if sigma is not None else self.scheduler.get_alpha_bar(t)
        xt = z / alpha_bar.sqrt()
        return xt.to(self.device, torch.float32).contiguous(), z.to(self.device)

    @ops.on_device
    def pred_xt(self, xt, t, sigma=None, batch_t=True):
        return self.model(self.convert_coordinate(xt, sigma, t), self._t_vec(t, len(xt), batch_t))

    @ops.on_device
    def encode_xt(self, xt, t, sigma=None, batch_t=True):
        return self.model.encode(self.convert_coordinate(xt, sigma, t), self._t_vec(t, len(xt), batch_t))

    @ops.on_device
    def forward_and_encode_xt(self, xt, t, sigma=None, batch_t=True):
        return self.model.forward_and_encode(self.convert_coordinate(xt, sigma, t), self._t_vec(t, len(xt), batch_t))

    def _t_vec(self, t, n, batch_t):
        t = torch.as_tensor(t)
        if batch_t:
            t = torch.ones(n) * t
        return t.reshape(-1).float()

    # ---- the device loop ----------------------------------------------------------------------------
    def _state(self, B):
        dev = self.device
        st = getattr(self, "_dev_state", None)
        if st is None or st["B"] != B or st["dev"] != dev:
            f = lambda: torch.empty(B, device=dev, dtype=torch.float32)
            st = dict(B=B, dev=dev, sigma_t=f(), sigma_prev=f(), t=f(), c_in=f(),
                      nan=torch.zeros(1, device=dev, dtype=torch.int32))
            self._dev_state = st
        return st

    def _nlc_step(self, xt, t_sched, sigma_sched, sigma_prev_sched, style, norm_eps, refine, per_sample=False,
                  prev_is_ratio=False, chunk_size=1):
        """get_denoise_vector (src/experiments.py:399-460) on the device.  Returns (eps_out, eps_sumsq).
        ``per_sample``: the scheduled sigma_t / t are the per-sample vectors already in the device state
        (projection_loop, image_sample.py:461-497); ``prev_is_ratio``: sigma_prev = sigma_t * sigma_prev_sched."""
        S = self.scheduler
        B = xt.shape[0]
        st = self._state(B)
        slopes = S.device_t_slopes(self.device)
        sumsq = ops.row_sumsq(xt) if refine else None
        ops.refine_sigma(sumsq, math.sqrt(self.dim), self.norm_max, self.norm_min, float(sigma_sched),
                         float(sigma_prev_sched), refine, S.device_sigmas(self.device) if refine else None,
                         float(t_sched), self.time_shift, st["sigma_t"], st["sigma_prev"], st["t"], st["c_in"],
                         sigma_in=st["sigma_t"] if per_sample else None, t_in=st["t"] if per_sample else None,
                         prev_is_ratio=prev_is_ratio, t_slopes=slopes if refine else None)
        if "pred" in style:
            if self.sigma_model is None:
                raise NlcError("style '%s' needs a sigma model (set_model)" % style)
            feat = self.model.run(xt, st["t"], mode="encode", in_scale=st["c_in"], feat_nhwc=True)
            if feat.dtype != self.sigma_model.compute_dtype:
                # the two networks may run in different precisions, as upstream (use_fp16 / use_sigma_fp16, image_sample.py:378-381):
                # the small feature map goes through the reference's own f32 NCHW format
                feat = ops.nchw_f32_to_nhwc(ops.nhwc_to_nchw_f32(feat), self.sigma_model.compute_dtype)
            r = self.sigma_model.run_nhwc(feat)
            ops.sigma_correct(r, style != "pred", S.device_sigmas(self.device), st["sigma_t"], st["sigma_prev"], st["t"],
                              st["c_in"], t_slopes=slopes)
        # the reference evaluates the eps network on len(xt) // chunk_size samples at a time (src/experiments.py:436-450: memory
        # only - no op of the networks couples samples, so the result is the same up to the f32 summation order of the kernels
        # that the per-launch batch selects (split-K, attention DMA form, pointwise tile thresholds): tests/test_loop_gpu.py gates
        # 1e-4.  chunk_size > 1 also shrinks every launch of the eps network: the CLIs and bench.py pass 1)
        micro = max(B // max(int(chunk_size), 1), 1)
        if micro >= B:
            eps_out = self.model.run(xt, st["t"], mode="forward", in_scale=st["c_in"])
        else:
            eps_out = torch.cat([self.model.run(xt[i:i + micro], st["t"][i:i + micro], mode="forward", in_scale=st["c_in"][i:i + micro]).clone()
                                 for i in range(0, B, micro)])
        C = self.data_shape[0]
        if self.learn_epsvar and eps_out.shape[1] != 2 * C:
            raise NlcError("learn_epsvar expects a 2C-channel network output")
        es = ops.row_sumsq(eps_out, d_used=self.dim) if norm_eps else None
        return eps_out, es

    def _sched_update(self, xt, eps_out, es, ind, constrain_fn, use_constraint, return_log, noise_list):
        """pred_xstart -> clip -> (constraint) -> pred_xprev (src/experiments.py:360-370) as fused kernels.
        Returns (x0_hat, x0, x_prev, eps_used)."""
        S = self.scheduler
        dev = self.device
        B, C = xt.shape[0], xt.shape[1]
        HW = xt.numel() // (B * C)
        st = self._state(B)
        stochastic = S.eta > 0 or S.variant in ("ddpm", "ddpm_orig")
        noise = None
        if stochastic:
            if noise_list is None:
                self.host_draws_used = getattr(self, "host_draws_used", 0) + 1
            noise = (noise_list[ind] if noise_list is not None else torch.randn(xt.shape)).to(dev, torch.float32).contiguous()
        x0 = torch.empty_like(xt)
        x_prev = torch.empty_like(xt)
        eps_used = torch.empty_like(xt) if return_log else None
        var_mode = VAR_MODES[S.sampler_var]
        if var_mode == VAR_MODES["learned"] and not self.learn_epsvar:
            raise NlcError("sampler_var 'learned' needs a network with learned variance (learn_epsvar)")
        d = SchedDesc(xt=xt.data_ptr(), eps_out=eps_out.data_ptr(), noise=None if noise is None else noise.data_ptr(),
                      sigma_t=st["sigma_t"].data_ptr(), sigma_prev=st["sigma_prev"].data_ptr(),
                      eps_norm_sumsq=None if es is None else es.data_ptr(), x0=x0.data_ptr(), x_prev=x_prev.data_ptr(),
                      eps_used=None if eps_used is None else eps_used.data_ptr(), B=B, C=C, Cnet=eps_out.shape[1], HW=HW,
                      variant=SCHED_VARIANTS[S.variant], clip=CLIP_MODES[self.clip_kind], var_mode=var_mode, phases=0,
                      eta=float(S.eta), min_var_coef=float(S.min_var_coef))
        ops.sched_x0(d)
        if self.clip_kind == "dynamic":
            dyn = ops.dynamic_threshold(x0, 0.99, 100.0)
            d.dyn_s = dyn.data_ptr()
        fused_mask = use_constraint and not return_log and hasattr(constrain_fn, "mask_chw") and hasattr(constrain_fn, "known")
        if fused_mask:
            # inpainting projection x0 - A^+(A x0 - y) == "copy the known pixels": one fused kernel does
            # clip + projection + x_prev
            d.mask, d.known = constrain_fn.mask_chw.data_ptr(), constrain_fn.known.data_ptr()
            ops.sched_step(d, st["nan"])
            x0_hat = x0
        elif use_constraint or (return_log and constrain_fn is not None):
            d.phases = 1                                   # clip only
            ops.sched_step(d)
            x0_hat = x0.clone() if return_log else x0
            if use_constraint:
                x0 = constrain_fn(x0).to(dev, torch.float32).contiguous()
                d.x0 = x0.data_ptr()
            d.phases = 2
            ops.sched_step(d, st["nan"])
        else:
            ops.sched_step(d, st["nan"])
            x0_hat = x0
        return x0_hat, x0, x_prev, eps_used

    @torch.no_grad()
    @ops.on_device
    def get_denoise_vector(self, xt, t, sigma_t, sigma_prev, style="base", norm_eps=False, refine_prior_sigma=False,
                           chunk_size=2):
        """Reference-shaped wrapper: returns (eps_mean, eps_logvar, sigma_t, sigma_prev) as (B,..) GPU tensors."""
        xt = xt.to(self.device, torch.float32).contiguous()
        B, C = xt.shape[0], self.data_shape[0]
        eps_out, es = self._nlc_step(xt, float(t), float(sigma_t), float(sigma_prev), style, norm_eps, refine_prior_sigma,
                                     chunk_size=chunk_size)
        st = self._state(B)
        eps = eps_out[:, :C].contiguous()
        if es is not None:
            eps = ops.scale_rows(eps, math.sqrt(self.dim) / torch.clamp(es.sqrt(), min=1e-12))
        sig_t, sig_p = st["sigma_t"].view(B, 1, 1, 1).clone(), st["sigma_prev"].view(B, 1, 1, 1).clone()
        learned = eps_out[:, C:] if self.learn_epsvar else None
        return eps, self.scheduler.get_eps_logvar(sig_t, sig_p, learned), sig_t, sig_p

    @torch.no_grad()
    @ops.on_device
    def denoise_loop(self, shape, gen=None, norm_init_noise=False, style="base", constrain_fn=None, norm_eps=False,
                     refine_prior_sigma=False, xT=None, return_log=True, chunk_size=2, sigma_pred_threshold=1000,
                     new_eta=None, constrain_loss=None, return_best=True, free_const_steps=-1, noise_list=None,
                     return_on_device=False, max_steps=None, start_step=0):
        """src/experiments.py:329-397.  ``noise_list`` (optional) supplies the per-step N(0,1) draws that
        stochastic samplers consume; by default they come from the global CPU generator, one
        ``randn(shape)`` per step, in step order.  ``return_on_device`` keeps the returned sample in HBM (the
        reference returns it on the CPU, :396; the sharded driver gathers it over RCCL first).  ``max_steps`` stops after
        that many timesteps of the schedule (parity checks against a partial oracle trajectory); with ``return_log`` the
        NLC-corrected per-sample sigma_t of every timestep is kept in ``self.sigma_trace`` (a list of CPU [B] tensors) and the
        state x_t every timestep STARTED from in ``self.xt_trace`` (CPU; teacher-forced per-timestep comparisons of two
        precisions feed one run's x_t to the other's ``get_denoise_vector``).  ``start_step`` = k resumes the schedule at its
        k-th timestep from ``xT`` = the state that timestep starts from (a run whose first k timesteps were made in another
        precision: tools/parity_trace.py --mixed)."""
        S = self.scheduler
        S.reset_state()
        dev = self.device
        sig_host = S.sampling_sigmas.detach().cpu()
        ts_host = S.timesteps.detach().cpu()
        self.host_draws_used = 0                 # randn(batch_shape) draws this call took from the host generator (sharded drivers
        if xT is None:                           # pad an early NaN break up to host_draws_per_batch(): image_sample.py)
            xt, zt = self.get_noise_xt(shape=shape, gen=gen, norm_noise=norm_init_noise, sigma=sig_host[0])
            self.host_draws_used = 1
        else:
            xt = xT.to(dev, torch.float32).contiguous()
            zt = self.convert_coordinate(xt, sigma=sig_host[0]) if return_log else None
        B = xt.shape[0]
        st = self._state(B)
        z_list, eps_list, x0_prec_list, x0_postc_list, const_loss_list = [], [], [], [], []
        if return_log:
            z_list = [zt.cpu()]
        eta0 = S.eta
        best_val, best_x0 = 10000, xt
        x0 = xt
        st["nan"].zero_()
        # The reference reads torch.isnan(xt).any() on the host after every step (:389).  Without logging / constraint
        # losses nothing else needs the host, so that read is taken one step late: step i's (sticky) flag is copied to
        # pinned memory behind step i, and looked at after step i+1 has been queued - the GPU never drains.  A NaN at
        # step i then discards the already-queued step i+1 and returns what the reference's break returns.
        lazy_nan = self.check_nan and not return_log and constrain_loss is None
        if lazy_nan:
            flag_host = torch.zeros(2, dtype=torch.int32).pin_memory()
            flag_ev = [torch.cuda.Event(), torch.cuda.Event()]
            x0_before = x0
        self.sigma_trace, self.xt_trace = [], []
        for ind, (t, t_prev) in enumerate(pairwise(ts_host.tolist())):
            if ind < start_step:
                S.i += 1
                continue
            if max_steps is not None and ind >= max_steps:
                break
            if return_log:
                self.xt_trace.append(xt.cpu())
            if ind == S.num_inference_steps - 1 and new_eta is not None:
                S.eta = new_eta
            cur_style, cur_refine = style, bool(refine_prior_sigma)
            if t > sigma_pred_threshold:
                cur_style, cur_refine = "base", False
            eps_out, es = self._nlc_step(xt, t, sig_host[ind], sig_host[ind + 1], cur_style, bool(norm_eps), cur_refine,
                                         chunk_size=chunk_size)
            use_constraint = constrain_fn is not None and (free_const_steps <= 0 or ind <= free_const_steps)
            x0_hat, x0, x_prev, eps_used = self._sched_update(xt, eps_out, es, ind, constrain_fn, use_constraint, return_log,
                                                             noise_list)
            S.i += 1
            xt = x_prev
            if constrain_loss is not None:
                const, _ = constrain_loss(x0.clamp(-1, 1))
                const_val = torch.mean(const)
                if const_val < best_val:
                    best_x0, best_val = x0, const_val
                if return_log:
                    const_loss_list.append(const.cpu())
            else:
                best_x0 = x0
            if return_log:
                self.sigma_trace.append(st["sigma_t"].cpu())
                z_list.append(ops.scale_rows(xt, torch.sqrt(1 / (st["sigma_prev"] ** 2 + 1))).cpu())   # logging only
                eps_list.append(eps_used.cpu())
                x0_prec_list.append(x0_hat.cpu())
                x0_postc_list.append(x0.cpu())
            if lazy_nan:
                flag_host[ind & 1:(ind & 1) + 1].copy_(st["nan"], non_blocking=True)
                flag_ev[ind & 1].record()
                if ind > 0:
                    flag_ev[(ind - 1) & 1].synchronize()
                    if int(flag_host[(ind - 1) & 1]) != 0:        # NaN at step ind-1: the reference stopped there
                        x0 = best_x0 = x0_before
                        break
                x0_before = x0
            elif self.check_nan and int(st["nan"].item()) != 0:       # torch.isnan(xt).any() -> break (:389)
                break
        S.eta = eta0
        out = best_x0 if return_best else x0
        if not return_on_device:
            out = out.cpu()
        return out, [z_list, eps_list, x0_prec_list, x0_postc_list, const_loss_list]

    @torch.no_grad()
    @ops.on_device
    def projection_loop(self, shape, gen=None, norm_init_noise=False, style="base", constrain_fn=None, norm_eps=False,
                        refine_prior_sigma=False, xT=None, return_log=False, chunk_size=2, sigma_estimate_rate=(1, 0, 0, 0),
                        constrain_loss=None, stop_condition=0.0, max_T=None, sigma_pred_threshold=1000, new_eta=None,
                        recal_sigma_prev=False, noise_list=None):
        """The 'project' sampling mode (image_sample.py:431-519): denoise_loop whose next (sigma_t, t) are
        re-estimated per sample from the norm of the new state (4-way blend ``sigma_estimate_rate``), with
        optional ``recal_sigma_prev`` and early stop on the constraint loss.  The per-sample sigma / t / last-norm
        state never leaves the device; the only per-step host read is ``t.max() > sigma_pred_threshold``, which
        the reference also does and which couples the batch (:471).
        Returns (x_cpu, [z_list, eps_list, x0_prec_list, x0_postc_list, sigma_list, const_loss_list])."""
        S = self.scheduler
        rate = [float(r) for r in sigma_estimate_rate]
        if len(rate) != 4:
            raise ValueError("sigma_estimate_rate needs 4 entries (image_sample.py:495 indexes [0..3])")
        if max_T is None:
            max_T = len(S.timesteps) - 1
        S.reset_state()
        dev = self.device
        sigs = S.sampling_sigmas.detach().cpu()
        ts_host = S.timesteps.detach().cpu()
        if xT is None:
            xt, zt = self.get_noise_xt(shape=shape, gen=gen, norm_noise=norm_init_noise, sigma=sigs[0])
        else:
            xt = xT.to(dev, torch.float32).contiguous()
            zt = self.convert_coordinate(xt, sigma=sigs[0]) if return_log else None
        B = xt.shape[0]
        st = self._state(B)
        x0 = xt
        eps_list, z_list, sigma_list, x0_prec_list, x0_postc_list, const_loss_list = [], [], [], [], [], []
        costheta = 0.99
        last_norm = ops.row_sumsq(xt).sqrt_().div_(math.sqrt(self.dim))
        if return_log:
            z_list = [zt.cpu()]
            sigma_list = [sigs[0].clone()]
        T = len(sigs)
        best_val, best_x0 = 10000, x0
        const_val = None
        sample_time_step = len(ts_host)
        eta0 = S.eta
        st["nan"].zero_()
        slopes = S.device_t_slopes(dev)
        nm = float(self.norm_max)
        for ind in range(max_T):
            if ind == sample_time_step - 1 and new_eta is not None:
                S.eta = new_eta
            sigma_prev_orig = sigs[-1] if ind >= T - 1 else sigs[ind + 1]
            per_sample = ind > 0
            if recal_sigma_prev:
                prev_arg, ratio = (sigs[ind + 1] / sigs[ind]), True          # sigma_prev = sigma_t * ratio (:463-464)
                if not per_sample:
                    prev_arg, ratio = sigs[0] * prev_arg, False
            else:
                prev_arg, ratio = sigma_prev_orig, False
            cur_style, cur_refine = style, bool(refine_prior_sigma)
            t_max = float(st["t"].max().item()) if per_sample else float(ts_host[0])
            if t_max > sigma_pred_threshold:
                cur_style, cur_refine = "base", False
            eps_out, es = self._nlc_step(xt, ts_host[0], sigs[0], prev_arg, cur_style, bool(norm_eps), cur_refine,
                                         per_sample=per_sample, prev_is_ratio=ratio, chunk_size=chunk_size)
            x0_hat, x0, x_prev, eps_used = self._sched_update(xt, eps_out, es, ind, constrain_fn, constrain_fn is not None,
                                                             return_log, noise_list)
            xt = x_prev
            if return_log:                                   # z uses the sigma_prev of THIS step: read before the update
                z_list.append(ops.scale_rows(xt, torch.sqrt(1 / (st["sigma_prev"] ** 2 + 1))).cpu())
            ops.proj_sigma(ops.row_sumsq(xt), math.sqrt(self.dim), nm, float(nm ** 2), costheta,
                           float(rate[0] * sigma_prev_orig), rate[1], rate[2], rate[3], S.device_sigmas(dev), slopes,
                           last_norm, st["sigma_t"], st["sigma_prev"], st["t"])
            if constrain_loss is not None:
                const, _ = constrain_loss(x0.clamp(-1, 1))
                const_val = torch.mean(const)
                if const_val < best_val:
                    best_x0, best_val = x0, const_val
                if return_log:
                    const_loss_list.append(const.cpu())
            else:
                best_x0 = x0
            if return_log:
                eps_list.append(eps_used.cpu())
                x0_prec_list.append(x0_hat.cpu())
                x0_postc_list.append(x0.cpu())
                sigma_list.append(st["sigma_t"].view(B, 1, 1, 1).cpu())
            if (self.check_nan and int(st["nan"].item()) != 0) or (const_val is not None and const_val <= stop_condition):
                break
        S.eta = eta0
        return best_x0.cpu(), [z_list, eps_list, x0_prec_list, x0_postc_list, sigma_list, const_loss_list]

    check_nan = True

    # ---- sigma-net training: the frozen-encoder half of one iteration -------------------------------
    @torch.no_grad()
    @ops.on_device
    def sigma_training_batch(self, batch_x, t, new_noise, microbatch=16):
        """What ``ImageExperiment.train`` computes per iteration BEFORE the sigma net sees anything
        (src/experiments.py:665-681): the regression target  dist_real = ||new_noise|| / sqrt(dim)  (B,1,1,1), the noised
        batch  noisy_x = scheduler.diffusion(batch_x, t, new_noise)  and the frozen eps model's features
        ``feat = cat_i model.encode(noisy_x[i : i + microbatch], t[i : i + microbatch])`` (NCHW f32).  The sigma net's own
        forward / backward / optimizer stay with the training framework; the cross-GPU half of :645-652 is
        ``sync_sigma_parameters`` / ``average_sigma_gradients`` below."""
        dev = self.device
        batch_x = batch_x.to(dev, torch.float32).contiguous()
        new_noise = new_noise.to(dev, torch.float32).contiguous()
        t = torch.as_tensor(t).long().reshape(-1)
        B = batch_x.shape[0]
        dist_real = (ops.row_sumsq(new_noise).sqrt() / math.sqrt(self.dim)).view(B, *([1] * (batch_x.dim() - 1)))
        noisy_x, _ = self.scheduler.diffusion(batch_x, t, new_noise)
        feats = [self.model.encode(noisy_x[i:i + microbatch], t[i:i + microbatch]) for i in range(0, B, microbatch)]
        return torch.cat(feats), dist_real, noisy_x


    # the two collectives DistributedDataParallel(sigma_model, bucket_cap_mb=128, broadcast_buffers=False) contributes to
    # ImageExperiment.train (src/experiments.py:645-652,682-686), for a training framework that keeps the sigma net's trainable
    # copy as ordinary tensors (one process per GPU, shard.init_from_env)
    @staticmethod
    def sync_sigma_parameters(params, src=0, bucket_mb=128):
        """DDP construction: rank ``src``'s parameters to every rank, in place, in ``bucket_mb`` buckets."""
        from . import shard
        shard.broadcast_parameters_(params, src=src, bucket_bytes=bucket_mb << 20)

    @staticmethod
    def average_sigma_gradients(grads, bucket_mb=128):
        """DDP's gradient exchange: mean over ranks, in place, one all-reduce per ``bucket_mb`` bucket.  (Upstream runs every
        forward under ``no_sync()`` (:682-686), so its own loop never reaches this; a loop that wants synchronous data-parallel
        training calls it between backward and the optimizer step.)"""
        from . import shard
        shard.allreduce_mean_(grads, bucket_bytes=bucket_mb << 20)


class ImageExperiment(ExperimentDiffusion):
    """src/experiments.py:553-560 (sampling; of the sigma net's training the frozen-encoder half and the cross-GPU exchange)."""

    def __init__(self, model, scheduler, batch_size=64, data_shape=(3, 32, 32), seed=0, device="cuda:0", save_folder="./",
                 dist_train=False, time_shift=0):
        super().__init__(model=model, scheduler=scheduler, batch_size=batch_size, data_shape=data_shape,
                         save_folder=save_folder, seed=seed, device=device, dist_train=dist_train, time_shift=time_shift)


from .edm_experiment import EDMImageExperiment  # noqa: E402,F401  (bottom import: edm_experiment needs ImageExperiment)
