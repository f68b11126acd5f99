#!/usr/bin/env python3
"""Headline benchmark: images/sec, 256x256 50-step DDIM + NLC on the ADM-256 UNet (BASELINE.json
configs[1]: batch 16 per GPU, bf16 operands / f32 accumulate, synthetic inputs and filler weights).

    python bench.py --gpus N --steps K --warmup W        # N > 1: starts its own N ranks (see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W           # or under a launcher that already set RANK / WORLD_SIZE

One "step" = one full sampling pass of one batch per GPU (ADM-256: 50 DDIM+NLC timesteps of 16 images):
per timestep  UNet.encode -> sigma net -> UNet.forward -> scheduler update.  Every rank samples its
own batches (weak scaling, independent samples); the finished samples stay in HBM and the
single collective is one RCCL all-gather of them inside the timed region.  Rank 0 prints ONE JSON line.

The timed region runs the product path with no instrumentation.  After it, rank 0 runs ONE more step with
HIP events around every convolution launch (on the launch stream) for the `roofline` object:
  roofline     all nlc_conv2d launches of a step: algorithmic direct-conv FLOPs / their summed duration;
               peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md).  `traffic` comes from separate rocprofv3
               --pmc passes of the same build (profiles/, tagged with the hash of the kernel sources).
  cpu_baseline the CPU oracle (oracle/, a port of the reference's PyTorch-CPU path) timed on the host
               cores of this box on a bounded sample of the same workload, extrapolated.

Other workloads (BASELINE.json configs[2], configs[3]; not the headline metric):  --config edm32 | celebahq256.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ADM256 = dict(image_size=256, num_channels=256, num_res_blocks=2, channel_mult="", learn_sigma=True,
              attention_resolutions="32,16,8", num_heads=4, num_head_channels=64, use_scale_shift_norm=True,
              resblock_updown=True, use_new_attention_order=False, sigma_block=2)
EDM32 = dict(img_resolution=32, in_channels=3, out_channels=3, augment_dim=9, model_channels=128, channel_mult=[2, 2, 2],
             num_blocks=4, attn_resolutions=[16], dropout=0.0, sigma_block=2, sigma_dropout=0.0)
CELEBAHQ = dict(ch=128, out_ch=3, ch_mult=[1, 1, 2, 2, 4, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0, in_channels=3,
                resamp_with_conv=True, feat_layer=1, type="simple", sigma_block=2, sigma_dropout=0.0)
# per image, one network evaluation with NLC: forward + encode + sigma net (BASELINE.md §2, 2*MAC of conv/linear/bmm)
GF_PER_IMAGE_STEP = 2239.67 + 580.29 + 3.95          # ADM-256
GF_EDM_EVAL = 42.38 + 13.76 + 0.25                    # EDM CIFAR-32 SongUNet
GF_CELEBA_STEP = 497.03 + 135.15 + 0.99               # CelebA-HQ-256 simple UNet
SIGMA_OVERRIDES = {"final_mlp.weight": 0.1, "final_mlp.bias": 0.5}


# ------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------
# --dtype -> (storage / operand type, matrix mode of f32 convolutions)
PRECISIONS = {"bf16": (torch.bfloat16, "native"), "f16": (torch.float16, "native"), "f32": (torch.float32, "native"),
              "f32x3": (torch.float32, "f16x3")}
# dense matrix peak the roofline fraction is priced against (MI355X_MICROARCH.md): 16-bit MFMA 2.5 PF; exact f32 MFMA 157.3 TF;
# f32x3 = three 16-bit passes per product, so 2.5 PF / 3 of ALGORITHMIC FLOPs is its ceiling
PEAKS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3, "f32x3": 2500.0 / 3}


def set_precision(module, prec):
    dtype, matmul = prec
    module.set_compute_dtype(dtype)
    module.set_matmul(matmul)
    return module


def build_models(cfg, device, prec):
    from diffusion_nlc_amd import script_util
    from diffusion_nlc_amd.filler import fill_state_dict
    eps, sig, fshape = script_util.create_sigma_eps_model(**cfg)
    eps.load_state_dict(fill_state_dict(eps.state_dict(), seed=0))
    sig.load_state_dict(fill_state_dict(sig.state_dict(), seed=1, overrides=SIGMA_OVERRIDES))
    set_precision(eps.to(device), prec)
    set_precision(sig.to(device), prec)
    return eps, sig


def make_experiment(cfg, device, dtype, batch, timesteps):
    from diffusion_nlc_amd.experiments import ImageExperiment
    from diffusion_nlc_amd.schedulers import get_sampler
    eps, sig = build_models(cfg, device, dtype)
    res = cfg["image_size"]
    sched = get_sampler("ddim", 1000, timesteps, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="learned", eta=0.0)
    sched.to(device)
    exp = ImageExperiment(eps, sched, batch_size=batch, data_shape=(3, res, res), seed=1234, device=device)
    exp.set_model(eps, sig, learn_epsvar=True)
    exp.set_norm_maxmin(0.0, 440.0 * res / 256)          # imagenet preset (image_sample.py:153-161)
    exp.set_clip_fn("dynamic")
    return exp


class AdmWorkload:
    """BASELINE.json configs[1]: ADM UNet 256x256, 50-step DDIM+NLC, batch 16 per GPU (the headline metric)."""
    name = "adm256"

    def __init__(self, args, device, dtype):            # dtype: a PRECISIONS entry
        self.cfg = dict(ADM256)
        if args.tiny:
            self.cfg.update(image_size=64, num_channels=64, channel_mult="1,2,2,4", attention_resolutions="16,8", num_head_channels=32)
        self.res, self.batch, self.timesteps, self.device = self.cfg["image_size"], args.batch or 16, args.timesteps or 50, device
        self.shape = (self.batch, 3, self.res, self.res)
        self.exp = None if args.dry_run else make_experiment(self.cfg, device, dtype, self.batch, self.timesteps)
        self.headline = self.res == 256 and self.timesteps == 50
        self.metric = ("images/sec whole-node, 256x256 50-step DDIM+NLC" if self.headline
                       else f"images/sec, {self.res}x{self.res} {self.timesteps}-step DDIM+NLC (debug configuration)")
        # (kept under 120 characters: the driver's record cuts longer strings)
        self.workload = (f"ADM UNet {self.res}x{self.res} (src/unet_adm.py), {self.timesteps}-step DDIM+NLC, batch {self.batch}/GPU, {args.dtype}, "
                         "dynamic clip, learned var, filler weights")
        self.gflop_per_image = self.timesteps * GF_PER_IMAGE_STEP if self.res == 256 else None

    @staticmethod
    def host_noise_spec(args):
        """(batch shape, seed) of the host draws, known before the models exist: main() replays the reference's generator on a
        thread while the networks are built and packed."""
        res = 64 if args.tiny else 256
        return (args.batch or 16, 3, res, res), 1234

    def inputs(self, n_total, world, rank, zs=None):
        from diffusion_nlc_amd import shard
        if zs is None:
            zs = shard.draw_initial_noise(self.shape, n_total, 1234, world, rank)      # host, reference draw order
        if self.exp is None:
            return [z.to(self.device) for z in zs]
        sigma0 = self.exp.scheduler.sampling_sigmas[0]
        return [(z / (1 / (sigma0 ** 2 + 1)).sqrt()).to(self.device) for z in zs]       # resident in HBM before timing

    def run(self, xT):
        x, _ = self.exp.denoise_loop(shape=self.shape, xT=xT, style="pred", norm_eps=True, refine_prior_sigma=True,
                                     return_log=False, chunk_size=1, sigma_pred_threshold=960, return_on_device=True)
        return x

    def cpu_baseline(self):
        """Time the oracle (CPU port) on DDIM+NLC timesteps of ADM-256 at B=1."""
        from diffusion_nlc_amd.filler import fill_state_dict
        from diffusion_nlc_amd.script_util import create_sigma_eps_model
        from oracle import adm
        from oracle.loop import DiffusionOracle
        from oracle.sched import get_sampler
        cfg = self.cfg
        cores = _host_cores()
        torch.set_num_threads(cores)
        ucfg, scfg, _ = adm.configs_from_factory(**cfg)
        eps_m, sig_m, _ = create_sigma_eps_model(**cfg)
        sd_e = fill_state_dict(eps_m.state_dict(), seed=0)
        sd_s = fill_state_dict(sig_m.state_dict(), seed=1, overrides=SIGMA_OVERRIDES)
        s = get_sampler("ddim", 1000, 50, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="learned", eta=0.0)
        res = cfg["image_size"]
        o = DiffusionOracle(lambda x, t: adm.unet(sd_e, ucfg, x, t, "forward"), lambda x, t: adm.unet(sd_e, ucfg, x, t, "encode"),
                            lambda f: adm.sigma_net(sd_s, scfg, f), s, (3, res, res), learn_epsvar=True, norm_min=0.0,
                            norm_max=440.0 * res / 256, clip_fn="dynamic")
        z = torch.randn((1, 3, res, res), generator=torch.Generator().manual_seed(1234))
        xt = z / (1 / (s.sampling_sigmas[0] ** 2 + 1)).sqrt()

        def one_timestep(x, ind):
            eps, lv, st, sp = o.get_denoise_vector(x, s.timesteps[ind], s.sampling_sigmas[ind], s.sampling_sigmas[ind + 1], "pred", True, True)
            x0 = o.clip(s.pred_xstart(x, eps, st))
            self.oracle_trace["x0"].append(x0.clone())
            self.oracle_trace["sigma"].append(st.clone())
            return s.pred_xprev(x0=x0, eps=eps, sigma_t=st, sigma_prev=sp, xt=x, log_variance=lv)

        # bounded sample: keep stepping the real trajectory until ~15 s of CPU work (at most 20 of the 50 timesteps)
        n, t0 = 0, time.perf_counter()
        self.oracle_trace = {"xT": xt.clone(), "x0": [], "sigma": []}       # for parity(): the SAME seeded image on the HIP path
        with torch.no_grad():
            while n < 20 and (n == 0 or time.perf_counter() - t0 < 15.0):
                xt = one_timestep(xt, n)
                n += 1
        dt = (time.perf_counter() - t0) / n
        return {"value": 1.0 / (50.0 * dt), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{n} DDIM+NLC timestep(s) of ADM-{res} at B=1 in f32 on the host ({dt:.2f} s each), extrapolated to 50 timesteps"}

    def parity(self, dtype_name):
        """BASELINE.json's metric is "images/sec ...; per-pixel L-inf vs CPU ref": the HIP path in the BENCHMARKED precision on the
        oracle leg's own seeded x_T (B = 1), compared timestep by timestep with what the oracle computed there.  `linf` / `rms`:
        the sample the loop would return after the last compared timestep (the clipped x0 estimate, in [-1, 1]); `sigma_rel`:
        the NLC-corrected sigma of that timestep.  A 16-bit run also reports the same comparison for `f32x3` - the precision of these
        kernels that carries the 1e-3 gate - under "f32x3" (same image, same timesteps)."""
        out = self._parity_one(dtype_name)
        if out is not None and dtype_name in ("bf16", "f16"):
            for m in (self.exp.model, self.exp.sigma_model):
                set_precision(m, PRECISIONS["f32x3"])
            try:
                ref = self._parity_one("f32x3")
            finally:
                for m in (self.exp.model, self.exp.sigma_model):
                    set_precision(m, PRECISIONS[dtype_name])
            if ref is not None:
                ref.pop("what", None)
                out["f32x3"] = ref
        return out

    def _parity_one(self, dtype_name):
        tr = getattr(self, "oracle_trace", None)
        if not tr or not tr["x0"]:
            return None
        n = len(tr["x0"])
        shape = (1, 3, self.res, self.res)
        exp = self.exp
        _, logs = exp.denoise_loop(shape=shape, xT=tr["xT"], style="pred", norm_eps=True, refine_prior_sigma=True, return_log=True,
                                   chunk_size=1, sigma_pred_threshold=960, max_steps=n)
        x0s, sig = logs[3], exp.sigma_trace
        per_step = [float((x0s[i].double() - tr["x0"][i].double()).abs().max()) for i in range(n)]
        d = x0s[n - 1].double() - tr["x0"][n - 1].double()
        srel = [float(((sig[i].double().view(-1) - tr["sigma"][i].double().view(-1)).abs() / tr["sigma"][i].double().view(-1)).max()) for i in range(n)]
        return {"dtype": dtype_name, "timesteps": n, "linf": per_step[-1], "rms": float(d.pow(2).mean().sqrt()),
                "sigma_rel": srel[-1], "linf_first_timestep": per_step[0], "sigma_rel_first_timestep": srel[0],
                "linf_max_over_timesteps": max(per_step),
                "sigma_rel_max_over_timesteps": max(srel),
                "what": "HIP path at B=1 on the cpu_baseline leg's seeded x_T vs the CPU oracle (f32), clipped x0 estimate after each of "
                        f"the first {n} DDIM+NLC timesteps; values are for timestep {n} unless named otherwise"}


class EdmWorkload:
    """BASELINE.json configs[2]: EDM SongUNet 32x32 (CIFAR-10 architecture), Heun sampler + NLC, 18 sigma steps
    (35 network evaluations with NLC), float64 state, batch 200 (the reference CLI default, edm_image_sample.py:35)."""
    name = "edm32"

    def __init__(self, args, device, dtype):
        from diffusion_nlc_amd import script_util
        from diffusion_nlc_amd.experiments import EDMImageExperiment
        from diffusion_nlc_amd.filler import fill_state_dict
        self.batch, self.steps, self.device = args.batch or 200, args.timesteps or 18, device
        self.shape = (self.batch, 3, 32, 32)
        self.exp = None
        if not args.dry_run:
            eps, sig, _ = script_util.create_edm_sigma_eps_model(**EDM32)
            tmpl = eps.state_dict()
            for k in tmpl:
                if k.endswith("resample_filter"):
                    tmpl[k] = torch.ones_like(tmpl[k]) / 4.0
            eps.load_state_dict(fill_state_dict(tmpl, seed=0))
            sig.load_state_dict(fill_state_dict(sig.state_dict(), seed=1, overrides=SIGMA_OVERRIDES))
            set_precision(eps.to(device), dtype)
            set_precision(sig.to(device), dtype)
            self.exp = EDMImageExperiment(eps, None, batch_size=self.batch, data_shape=(3, 32, 32), seed=0, device=device,
                                          num_timesteps=self.steps)
            self.exp.set_model(eps, sig, learn_epsvar=False)
            self.exp.set_norm_maxmin(0.0, 54.63)
        self.headline = False
        self.metric = f"images/sec, EDM 32x32 Heun+NLC, {self.steps} sigma steps"
        self.workload = (f"EDM SongUNet 32x32 (src/edm_networks.py, 128 ch, mult 2-2-2), Heun + NLC 'pred_partial,pred', {self.steps} sigma steps "
                         f"({2 * self.steps - 1} evaluations), f64 state, batch {self.batch} per GPU, {args.dtype}, filler weights")
        self.gflop_per_image = (2 * self.steps - 1) * GF_EDM_EVAL

    host_noise_spec = None                      # per-sample generators: each rank draws only its own samples

    def inputs(self, n_total, world, rank, zs=None):
        from diffusion_nlc_amd import shard
        from diffusion_nlc_amd.experiments import StackedRandomGenerator
        out = []
        for j in shard.owned_batches(n_total, world, rank):                  # per-sample host generators, seeds = global sample index
            g = StackedRandomGenerator(self.device, range(j * self.batch, (j + 1) * self.batch))
            out.append(g.randn(self.shape))
        return out

    def run(self, lat):
        return self.exp.edm_sampler(shape=self.shape, latents=lat, style="pred_partial,pred", norm_eps="000", eps_ratio=0.5,
                                    eps_scale=1.0, use_second_order=True)

    def cpu_baseline(self):
        from diffusion_nlc_amd import script_util
        from diffusion_nlc_amd.filler import fill_state_dict
        from oracle import edm
        from oracle.loop import EdmOracle
        torch.set_num_threads(_host_cores())
        eps, sig, _ = script_util.create_edm_sigma_eps_model(**EDM32)
        tmpl = eps.state_dict()
        for k in tmpl:
            if k.endswith("resample_filter"):
                tmpl[k] = torch.ones_like(tmpl[k]) / 4.0
        sd_e = fill_state_dict(tmpl, seed=0)
        sd_s = fill_state_dict(sig.state_dict(), seed=1, overrides=SIGMA_OVERRIDES)
        cfg = edm.EdmConfig(img_resolution=32, in_channels=3, out_channels=3, augment_dim=9, model_channels=128, channel_mult=(2, 2, 2),
                            num_blocks=4, attn_resolutions=(16,), sigma_block=2)
        _, dim = edm.sigma_dims(cfg)
        B, steps = 16, 3
        o = EdmOracle(lambda x, t: edm.unet(sd_e, cfg, x, t, "forward"), lambda x, t: edm.unet(sd_e, cfg, x, t, "encode"),
                      lambda f: edm.sigma_net(sd_s, dim, cfg.sigma_block, f), (3, 32, 32), num_timesteps=steps, norm_min=0.0, norm_max=54.63)
        lat = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(0))
        n, t0 = 0, time.perf_counter()
        while n == 0 or (time.perf_counter() - t0 < 15.0 and n < 8):
            o.edm_sampler(lat, style="pred_partial,pred", norm_eps="000", eps_ratio=0.5, eps_scale=1.0, use_second_order=True)
            n += 1
        dt = (time.perf_counter() - t0) / n / (2 * steps - 1) / B             # s per image per evaluation
        return {"value": 1.0 / ((2 * self.steps - 1) * dt), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{n} x ({steps}-step Heun+NLC = {2 * steps - 1} evaluations) at B={B} in f32/f64 on the host, extrapolated to {2 * self.steps - 1} evaluations"}


class CelebaWorkload:
    """BASELINE.json configs[3]: CelebA-HQ-256 'simple' UNet, inpainting restoration (seeded random 50 % mask,
    functions/svd_operators.py:324-359), 100-step DDIM+NLC, batch 8; the projection is fused into nlc_sched_step."""
    name = "celebahq256"

    def __init__(self, args, device, dtype):
        from diffusion_nlc_amd import script_util
        from diffusion_nlc_amd.constraint_functions import Constraint_Function, Inpainting
        from diffusion_nlc_amd.experiments import ImageExperiment
        from diffusion_nlc_amd.filler import fill_state_dict
        from diffusion_nlc_amd.schedulers import get_sampler
        ns = argparse.Namespace
        self.batch, self.timesteps, self.device, self.res = args.batch or 8, args.timesteps or 100, device, 256
        self.shape = (self.batch, 3, 256, 256)
        self.exp = None
        if not args.dry_run:
            config = ns(model=ns(**CELEBAHQ), data=ns(image_size=256), diffusion=ns(num_diffusion_timesteps=1000))
            eps, sig, _ = script_util.create_simple_sigma_eps_model(config)
            eps.load_state_dict(fill_state_dict(eps.state_dict(), seed=0))
            sig.load_state_dict(fill_state_dict(sig.state_dict(), seed=1, overrides=SIGMA_OVERRIDES))
            set_precision(eps.to(device), dtype)
            set_precision(sig.to(device), dtype)
            s = get_sampler("ddim", 1000, self.timesteps, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall", eta=0.0)
            s.to(device)
            self.exp = ImageExperiment(eps, s, batch_size=self.batch, data_shape=(3, 256, 256), seed=5, device=device)
            self.exp.set_model(eps, sig, learn_epsvar=False)
            self.exp.set_norm_maxmin(0.0, 397.0)                             # image_sample.py:171-174
            self.exp.set_clip_fn("clamp")
            g = torch.Generator().manual_seed(11)
            missing_r = torch.randperm(256 * 256, generator=g)[: 256 * 256 // 2].long() * 3
            missing = torch.cat([missing_r, missing_r + 1, missing_r + 2], dim=0)
            self.op = Inpainting(3, 256, missing, device)
            self.cf = Constraint_Function("inpainting_random", self.op, channels=3, image_size=256)
            x_gt = torch.rand(self.shape, generator=g) * 2 - 1
            self.bound = self.cf.bind(self.op.A(x_gt), self.shape)
        self.headline = False
        self.metric = f"images/sec, CelebA-HQ 256x256 inpainting, {self.timesteps}-step DDIM+NLC"
        self.workload = (f"simple UNet 256x256 (src/unet_simple.py, ch 128, mult 1-1-2-2-4-4), inpainting (random 50 % mask), {self.timesteps}-step "
                         f"DDIM+NLC, batch {self.batch} per GPU, {args.dtype}, clamp clip, fixedsmall variance, filler weights")
        self.gflop_per_image = self.timesteps * GF_CELEBA_STEP

    @staticmethod
    def host_noise_spec(args):
        return (args.batch or 8, 3, 256, 256), 5

    def inputs(self, n_total, world, rank, zs=None):
        from diffusion_nlc_amd import shard
        if zs is None:
            zs = shard.draw_initial_noise(self.shape, n_total, 5, world, rank)
        if self.exp is None:
            return [z.to(self.device) for z in zs]
        sigma0 = self.exp.scheduler.sampling_sigmas[0]
        return [(z / (1 / (sigma0 ** 2 + 1)).sqrt()).to(self.device) for z in zs]

    def run(self, xT):
        x, _ = self.exp.denoise_loop(shape=self.shape, xT=xT, style="pred", constrain_fn=self.bound, norm_eps=True, refine_prior_sigma=True,
                                     return_log=False, chunk_size=1, sigma_pred_threshold=960, return_on_device=True)
        return x

    def cpu_baseline(self):
        from diffusion_nlc_amd import script_util
        from diffusion_nlc_amd.filler import fill_state_dict
        from oracle import simple
        from oracle.loop import DiffusionOracle
        from oracle.sched import get_sampler
        torch.set_num_threads(_host_cores())
        ns = argparse.Namespace
        config = ns(model=ns(**CELEBAHQ), data=ns(image_size=256), diffusion=ns(num_diffusion_timesteps=1000))
        eps, sig, _ = script_util.create_simple_sigma_eps_model(config)
        sd_e = fill_state_dict(eps.state_dict(), seed=0)
        sd_s = fill_state_dict(sig.state_dict(), seed=1, overrides=SIGMA_OVERRIDES)
        cfg = simple.SimpleConfig(ch=128, out_ch=3, ch_mult=(1, 1, 2, 2, 4, 4), num_res_blocks=2, attn_resolutions=(16,), in_channels=3,
                                  resolution=256, resamp_with_conv=True, feat_layer=1, sigma_block=2)
        _, dim = simple.sigma_dims(cfg)
        s = get_sampler("ddim", 1000, self.timesteps, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall", eta=0.0)
        o = DiffusionOracle(lambda x, t: simple.unet(sd_e, cfg, x, t, "forward"), lambda x, t: simple.unet(sd_e, cfg, x, t, "encode"),
                            lambda f: simple.sigma_net(sd_s, dim, cfg.sigma_block, f), s, (3, 256, 256), learn_epsvar=False,
                            norm_min=0.0, norm_max=397.0, clip_fn="clamp")
        z = torch.randn((1, 3, 256, 256), generator=torch.Generator().manual_seed(5))
        xt = z / (1 / (s.sampling_sigmas[0] ** 2 + 1)).sqrt()
        n, t0 = 0, time.perf_counter()
        with torch.no_grad():
            while n < 20 and (n == 0 or time.perf_counter() - t0 < 15.0):
                eps_, lv, st, sp = o.get_denoise_vector(xt, s.timesteps[n], s.sampling_sigmas[n], s.sampling_sigmas[n + 1], "pred", True, True)
                x0 = o.clip(s.pred_xstart(xt, eps_, st))
                xt = s.pred_xprev(x0=x0, eps=eps_, sigma_t=st, sigma_prev=sp, xt=xt, log_variance=lv)
                n += 1
        dt = (time.perf_counter() - t0) / n
        return {"value": 1.0 / (self.timesteps * dt), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{n} DDIM+NLC timestep(s) of the CelebA-HQ-256 simple UNet at B=1 in f32 on the host ({dt:.2f} s each, projection not timed), "
                          f"extrapolated to {self.timesteps} timesteps"}


WORKLOADS = {"adm256": AdmWorkload, "edm32": EdmWorkload, "celebahq256": CelebaWorkload}

# BASELINE.json configs[0] / BASELINE.md §3b item 3: the reference's own CPU-runnable case - unet_simple 32x32, 10-step DDIM+NLC,
# batch 4 (image_sample.py --synthetic cifar_tiny; tests/golden/loop_simple_pred.npz is the reference's own run of it)
CFG0 = dict(ch=64, out_ch=3, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[16], dropout=0.0, in_channels=3,
            resamp_with_conv=True, feat_layer=1, type="simple", sigma_block=2, sigma_dropout=0.0)


def cfg0_baseline(device):
    """configs[0] run IN FULL on the host cores (the oracle = CPU port of the reference loop), then the same seeded x_T on the HIP path
    in f32: images/sec of the CPU run, and the per-pixel L-inf between the two final samples."""
    from diffusion_nlc_amd import script_util
    from diffusion_nlc_amd.experiments import ImageExperiment
    from diffusion_nlc_amd.filler import fill_state_dict
    from diffusion_nlc_amd.schedulers import get_sampler as hip_sampler
    from oracle import simple
    from oracle.loop import DiffusionOracle
    from oracle.sched import get_sampler
    torch.set_num_threads(_host_cores())
    ns = argparse.Namespace
    config = ns(model=ns(**CFG0), data=ns(image_size=32), diffusion=ns(num_diffusion_timesteps=1000))
    eps, sig, _ = script_util.create_simple_sigma_eps_model(config)
    sd_e = fill_state_dict(eps.state_dict(), seed=0)
    sd_s = fill_state_dict(sig.state_dict(), seed=1, overrides=SIGMA_OVERRIDES)
    cfg = simple.SimpleConfig(ch=64, out_ch=3, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(16,), in_channels=3,
                              resolution=32, resamp_with_conv=True, feat_layer=1, sigma_block=2)
    _, dim = simple.sigma_dims(cfg)
    B, steps, shape = 4, 10, (4, 3, 32, 32)
    kw = dict(sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall", eta=0.0)
    s = get_sampler("ddim", 1000, steps, **kw)
    o = DiffusionOracle(lambda x, t: simple.unet(sd_e, cfg, x, t, "forward"), lambda x, t: simple.unet(sd_e, cfg, x, t, "encode"),
                        lambda f: simple.sigma_net(sd_s, dim, cfg.sigma_block, f), s, (3, 32, 32), learn_epsvar=False,
                        norm_min=0.0, norm_max=54.63, clip_fn="clamp")
    z = torch.randn(shape, generator=torch.Generator().manual_seed(1234))
    xT = z / (1 / (s.sampling_sigmas[0] ** 2 + 1)).sqrt()
    run_cpu = lambda: o.denoise_loop(shape, style="pred", norm_eps=True, refine_prior_sigma=True, xT=xT.clone(), sigma_pred_threshold=960)
    run_cpu()                                                   # untimed: thread pool, allocator
    n, t0 = 0, time.perf_counter()
    while n == 0 or (time.perf_counter() - t0 < 6.0 and n < 50):
        x_cpu = run_cpu()
        n += 1
    dt = (time.perf_counter() - t0) / n
    out = {"value": B / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n} full runs of unet_simple 32x32 (ch 64, mult 1-2-2), 10-step DDIM+NLC, batch {B}, f32 on the host ({dt:.2f} s each); "
                     "nothing extrapolated"}
    eps.load_state_dict(sd_e)
    sig.load_state_dict(sd_s)
    eps.to(device)
    sig.to(device)
    hs = hip_sampler("ddim", 1000, steps, **kw)
    hs.to(device)
    exp = ImageExperiment(eps, hs, batch_size=B, data_shape=(3, 32, 32), seed=1234, device=device)
    exp.set_model(eps, sig, learn_epsvar=False)
    exp.set_norm_maxmin(0.0, 54.63)
    exp.set_clip_fn("clamp")
    run_hip = lambda: exp.denoise_loop(shape=shape, xT=xT.to(device), style="pred", norm_eps=True, refine_prior_sigma=True, return_log=False,
                                       chunk_size=1, sigma_pred_threshold=960, return_on_device=True)[0]
    x_hip = run_hip()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        x_hip = run_hip()
    torch.cuda.synchronize()
    out["hip_f32_images_per_sec"] = 5 * B / (time.perf_counter() - t0)
    out["hip_f32_linf_vs_cpu"] = float((x_hip.cpu().double() - x_cpu.double()).abs().max())
    return out


def _host_cores():
    # the GPU box shares its host: a 1-GPU slot owns 16 cores (os.cpu_count() reports the whole machine)
    return min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))


def csrc_sha16():
    """Hash of the convolution kernel sources: ties a PMC traffic profile to the build it was taken on."""
    h = hashlib.sha256()
    for f in sorted((ROOT / "diffusion-nlc_amd" / "csrc").glob("conv_*")):
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------------------
# launcher
# ------------------------------------------------------------------------------------------------------
def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) outside a launcher: start N ranks as CHILD processes - one per GPU, rendezvous on
    127.0.0.1 - and relay their output (rank 0 prints the JSON line).  The parent never touches the GPU and never
    re-execs itself; it only waits and returns the children's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC (RCCL between processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, _host_cores() // max(args.gpus, 1))))
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="adm256", choices=sorted(WORKLOADS), help="adm256 = the headline metric (BASELINE.json configs[1])")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (0 = the workload's default)")
    ap.add_argument("--timesteps", type=int, default=0, help="sampler timesteps (0 = the workload's default; ADM-256: 50 = the headline metric)")
    ap.add_argument("--dtype", default="bf16", choices=sorted(PRECISIONS),
                    help="bf16 (BASELINE.json configs[1]) | f16 (the reference's own use_fp16 mode) | f32 (exact f32 MFMA) | "
                         "f32x3 (f32 storage, split-f16 three-pass matrix math: the 1e-3-parity path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay each network evaluation from a captured hipGraph")
    ap.add_argument("--ops-flag", action="append", default=[], metavar="NAME=VALUE", help="A/B: set a module-level switch of diffusion_nlc_amd.ops")
    ap.add_argument("--gn-fusion", action="store_true", help="A/B: GroupNorm in the 3x3 convs' LDS prologue instead of a separate pass")
    ap.add_argument("--tiny", action="store_true", help="64x64 debugging configuration (NOT the headline metric)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / sharding / gather plumbing only: no sampling (runs without a GPU under gloo); never a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))                      # before anything in this process touches the GPU

    from diffusion_nlc_amd import ops, shard
    ops.FUSE_GN_CONV = bool(args.gn_fusion)
    for kv in args.ops_flag:                      # A/B switches of diffusion_nlc_amd.ops, e.g. --ops-flag FUSE_GN_POOL=0
        k, v = kv.split("=", 1)
        if not hasattr(ops, k):
            raise SystemExit(f"bench.py: ops has no switch {k!r}")
        setattr(ops, k, type(getattr(ops, k))(int(v)) if isinstance(getattr(ops, k), (bool, int)) else v)
    use_gpu = torch.cuda.is_available() and not (args.dry_run and os.environ.get("NLC_BENCH_FORCE_CPU"))
    if not use_gpu and not args.dry_run:
        raise SystemExit("bench.py measures the HIP path: it needs a GPU (only --dry-run runs without one)")
    rank, world, local = shard.init_from_env("nccl" if use_gpu else "gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    prec = PRECISIONS[args.dtype]
    dtype = prec[0]
    n_total = (args.warmup + args.steps) * world
    # Every rank replays the reference's single host generator over ALL global batches and keeps its own (shard.py): at
    # --gpus 8 --steps 20 that is 168 draws of 12.6 MB per rank.  Off the critical path: a thread draws while the networks
    # are built, filled and packed (torch.randn releases the GIL).
    spec = WORKLOADS[args.config].host_noise_spec
    noise_box, noise_thread, t_setup = {}, None, time.perf_counter()
    if spec is not None:
        import threading
        nshape, nseed = spec(args)
        noise_thread = threading.Thread(target=lambda: noise_box.update(zs=shard.draw_initial_noise(nshape, n_total, nseed, world, rank)))
        noise_thread.start()
    wl = WORKLOADS[args.config](args, device, prec)
    t_models = time.perf_counter() - t_setup
    if args.graph and wl.exp is not None:
        wl.exp.use_graphs = True
    if noise_thread is not None:
        noise_thread.join()
        assert tuple(noise_box["zs"][0].shape) == tuple(wl.shape)
    xs = wl.inputs(n_total, world, rank, noise_box.get("zs"))               # resident on the device before timing
    t_setup = time.perf_counter() - t_setup

    def one(x):
        return x * 0.5 if args.dry_run else wl.run(x)

    def barrier():
        if use_gpu:
            torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        if use_gpu:
            torch.cuda.synchronize()

    for i in range(args.warmup):
        one(xs[i])
    barrier()
    t0 = time.perf_counter()
    outs = [one(xs[args.warmup + i]) for i in range(args.steps)]           # finished samples stay on the device
    local_out = torch.stack(outs)
    if use_gpu and world > 1:
        torch.cuda.synchronize()            # phase stamp only (the collective below would wait for these kernels anyway)
    t1 = time.perf_counter()
    gathered = shard.gather_samples(local_out, args.steps * world, world, rank)     # the one all-gather (RCCL over xGMI)
    if use_gpu and world > 1:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    # [whole region, sampling, all-gather (includes waiting for the slowest rank to arrive), final barrier, setup, models]
    phases = torch.tensor([elapsed, t1 - t0, t2 - t1, t0 + elapsed - t2, t_setup, t_models], device=device, dtype=torch.float64)
    t = phases[:1].clone()
    per_rank = phases.view(1, -1)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        per_rank = phases.new_empty((world, phases.numel()))
        torch.distributed.all_gather_into_tensor(per_rank.view(-1), phases)
    elapsed = float(t.item())
    per_rank = per_rank.cpu()
    assert gathered.shape[0] == args.steps * world and torch.isfinite(gathered).all()

    images = wl.batch * args.steps * world
    line = {
        "metric": wl.metric if not args.dry_run else "DRY RUN (launcher / gather plumbing only, nothing sampled)",
        "value": images / elapsed, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic" if not args.dry_run else "none (dry run)",
        "config": {"workload": wl.workload, "global_batch": wl.batch * world,
                   "parallelism": f"dp{world} (independent samples, one all-gather)"},
    }
    if args.graph:
        line["config"]["hipgraph"] = True
    # what the first multi-GPU run needs to be diagnosable: the world size the process group really has, and where every rank's
    # time went (a slow rank shows up as the OTHER ranks' gather time)
    line["ranks_seen"] = torch.distributed.get_world_size() if (world > 1 and torch.distributed.is_initialized()) else 1
    line["per_rank_seconds"] = {k: [round(float(v), 4) for v in per_rank[:, i]] for i, k in
                                enumerate(["timed_region", "sampling", "all_gather", "final_barrier", "setup_untimed", "model_build_untimed"])}
    if rank == 0:
        if not args.dry_run and wl.gflop_per_image:
            line["end_to_end_tflops_per_gpu"] = (images / world) * wl.gflop_per_image / 1e3 / elapsed
        if not args.dry_run and not args.no_roofline:
            line["roofline"] = roofline_leg(wl, xs[0], dtype, args, line["ms_per_step"])
        if not args.dry_run and not args.no_cpu_baseline and not args.tiny and world == 1:
            line["cpu_baseline"] = wl.cpu_baseline()
            if wl.name == "adm256":
                line["cpu_baseline"]["configs0_unet_simple_32"] = cfg0_baseline(device)
            if hasattr(wl, "parity"):
                par = wl.parity(args.dtype)
                if par:
                    line["parity"] = par
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def roofline_leg(wl, x, dtype, args, ms_per_step):
    """One more step of the same workload (outside the timed region, rank 0 only, no collective) with HIP events on the
    launch stream around every nlc_conv2d launch."""
    from diffusion_nlc_amd import ops
    prof = []
    graphs = getattr(wl.exp, "use_graphs", False)
    wl.exp.use_graphs = False                       # events cannot be recorded per kernel inside a graph replay
    ops.CONV_PROFILE = prof
    aprof = ops.ATTN_PROFILE = []
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wl.run(x)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    finally:
        ops.CONV_PROFILE = None
        ops.ATTN_PROFILE = None
        wl.exp.use_graphs = graphs
    sel = [(e0.elapsed_time(e1), f, shp) for e0, e1, f, d, shp in prof if d == dtype]
    tot_ms, tot_fl, n = sum(m for m, _, _ in sel), sum(f for _, f, _ in sel), len(sel)
    ach = tot_fl / (tot_ms * 1e-3) / 1e12
    peak = PEAKS[args.dtype]
    out = {"bound": "mfma",
           "kernel": f"nlc_conv2d: conv_halo_kernel<{args.dtype}> (3x3, >= 64 tiles) + conv_small_kernel (3x3 on <= 32-wide maps, GroupNorm fused) + conv_fast_kernel<{args.dtype},9|1> + conv_pw(r)_kernel<{args.dtype}> (1x1, >= 768 tiles) + conv_igemm_kernel",
           "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None, "launches": n,
           "avg_launch_us": 1e3 * tot_ms / max(n, 1), "avg_launch_gflop": tot_fl / max(n, 1) / 1e9,
           # conv time of the instrumented step over the TIMED region's ms_per_step (reproducible from this line:
           # launches x avg_launch_us / ms_per_step); the instrumented step itself ran `instrumented_step_ms`
           "share_of_step": tot_ms / ms_per_step, "instrumented_step_ms": 1e3 * wall, "dominant_frac": None,
           "measured": "HIP events around every conv launch of one extra (untimed) step",
           "note": "the 1x1 skip projections of the ResBlocks also write act(GroupNorm(x)) of their input (nlc_conv_desc.norm_out): "
                   "their whole launch time counts here, the GroupNorm pass it replaces never did"}
    if aprof:
        # the attention launches of the same step (algorithmic QK^T + PV FLOPs; the matrix-pipe utilisation itself is a counter /
        # in-kernel-clock measurement: profiles/r05_summary.md section 3)
        ams = sum(e0.elapsed_time(e1) for e0, e1, _, _ in aprof)
        afl = sum(f for _, _, f, _ in aprof)
        big = [(e0.elapsed_time(e1), f) for e0, e1, f, shp in aprof if shp[0] >= 1024]
        out["attention"] = {"launches": len(aprof), "total_ms": ams, "tflops": afl / ams / 1e9, "frac_of_peak": afl / ams / 1e9 / peak,
                            "share_of_step": ams / ms_per_step}
        if big:
            out["attention"]["t1024_launch_us"] = 1e3 * sum(m for m, _ in big) / len(big)
            out["attention"]["t1024_tflops"] = sum(f for _, f in big) / sum(m for m, _ in big) / 1e9
    dom = []
    if wl.name == "adm256" and wl.res == 256:
        # the dominant launch's OWN fraction of peak (conv3x3 256->256 @256x256: the largest single share of the step)
        dom = [(m, f) for m, f, shp in sel if shp[:4] == (wl.batch * 256 * 256, 256, 9, 256) and shp[5] == 0]
    if dom:
        out["dominant_frac"] = sum(f for _, f in dom) / sum(m for m, _ in dom) / 1e9 / peak
        out["dominant_launch"] = {"shape": "conv3x3 256->256 @256x256, B=%d" % wl.batch, "launches": len(dom),
                                  "launch_us": 1e3 * sum(m for m, _ in dom) / len(dom),
                                  "tflops": sum(f for _, f in dom) / sum(m for m, _ in dom) / 1e9}
    # HBM bytes per launch from the separate rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE),
    # tagged with the hash of the kernel sources they were taken on
    tfiles = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))          # the newest round's profile
    tpath = tfiles[-1] if tfiles else None
    if wl.name == "adm256" and wl.res == 256 and args.dtype == "bf16" and tpath is not None:
        tj = json.loads(tpath.read_text())
        out["traffic"] = tj.get("conv2d_bytes_per_launch")
        out["traffic_source"] = {"file": f"profiles/{tpath.name}", "csrc_sha16": tj.get("csrc_sha16"),
                                 "matches_this_build": tj.get("csrc_sha16") == csrc_sha16()}
        if dom and "dominant" in tj:
            out["traffic_of_dominant_launch"] = {"kernel": tj["dominant"]["kernel"], "traffic": tj["dominant"]["traffic_bytes_per_launch"],
                                                 "algorithmic_bytes": tj["dominant"]["algorithmic_bytes_per_launch"],
                                                 "launch_us": 1e3 * sum(m for m, _ in dom) / len(dom),
                                                 "tflops": sum(f for _, f in dom) / sum(m for m, _ in dom) / 1e9}
    if os.environ.get("NLC_BENCH_SHAPES"):
        agg = {}
        for m, f, shp in sel:
            a = agg.setdefault(shp, [0, 0.0, 0.0])
            a[0] += 1; a[1] += m; a[2] += f
        for shp, (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
            print(f"# conv M={shp[0]} N={shp[1]} taps={shp[2]} Cin={shp[3]} s={shp[4]} ups={shp[5]} C1={shp[6]}: "
                  f"{cnt} launches, {ms:.1f} ms, {fl / ms / 1e9:.0f} TFLOP/s", file=sys.stderr)
    return out


if __name__ == "__main__":
    main()
