#!/usr/bin/env python3
"""DDIM-family sampling entry point with the reference's flags (drop-in for image_sample.py).

Same flag names, defaults and method presets as the reference (image_sample.py:32-96,143-268) and the
same call order in ``main`` (:712-860): build models -> load checkpoints -> sampler -> ImageExperiment
-> evaluate_unconstraint -> results.json.  Everything numeric runs on the HIP path of
diffusion-nlc_amd/.  Differences, all on side effects the reference hard-codes:

* FID (pytorch_fid + InceptionV3 download) and PNG writing are optional: missing packages/files are
  skipped, FID is reported as NaN.
* per-step logging (``return_log``, hard-coded True in the reference's main, :822) is the flag
  ``--return_log`` (default 0): the 2.5 GB-per-batch history is only copied to the host when asked for.
* ``--synthetic NAME`` (extension): run without ``store/`` / ``results/`` files, with a built-in model
  configuration and the deterministic filler weights (the reference ships neither configs nor checkpoints).
* ``--constraint``: the inpainting operators (``inpainting``, ``inpainting_half``) with ``--constraint_proj svd``
  run on the HIP path; the other SVD operators and the GD projections raise NotImplementedError.
  With ``--synthetic`` the validation images are seeded U(0,1) tensors and the mask is the seeded random 50 %
  of pixels of BASELINE config 4 (no dataset / ``store/inp_masks`` files ship with the reference).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import random
import shutil
import sys
from pathlib import Path
from time import time

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from src.experiments import ImageExperiment                                    # noqa: E402
from src.schedulers import get_sampler, redesign_sigma                          # noqa: E402
from src.script_util import create_sigma_eps_model, create_simple_sigma_eps_model   # noqa: E402
from src.utils import get_model_size                                            # noqa: E402

SYNTHETIC = {
    # name -> (yaml-equivalent config, dataset preset name)
    "imagenet256": (dict(model=dict(type="openai", image_size=256, num_channels=256, num_res_blocks=2, learn_sigma=True,
                                    attention_resolutions="32,16,8", num_head_channels=64, use_scale_shift_norm=True,
                                    resblock_updown=True, use_fp16=True, use_new_attention_order=False),
                         diffusion=dict(num_diffusion_timesteps=1000, beta_schedule="linear"),
                         data=dict(dataset="ImageNet", image_size=256, channels=3, subset_1k=False)), "imagenet"),
    "cifar_tiny": (dict(model=dict(type="simple", ch=64, out_ch=3, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[16],
                                   dropout=0.0, in_channels=3, resamp_with_conv=True, use_fp16=False),
                        diffusion=dict(num_diffusion_timesteps=1000, beta_schedule="linear"),
                        data=dict(dataset="CIFAR10", image_size=32, channels=3, subset_1k=False)), "celeba_hq"),
    # a small ADM (the reference's "openai" type) in its use_fp16 mode: the file-based path's stand-in in tests/test_cli_files_gpu.py
    "adm_tiny": (dict(model=dict(type="openai", image_size=64, num_channels=64, num_res_blocks=1, channel_mult="1,2,2,4", learn_sigma=True,
                                 attention_resolutions="16,8", num_head_channels=32, use_scale_shift_norm=True, resblock_updown=True,
                                 use_fp16=True, use_new_attention_order=False),
                      diffusion=dict(num_diffusion_timesteps=1000, beta_schedule="linear"),
                      data=dict(dataset="ImageNet", image_size=64, channels=3, subset_1k=False)), "imagenet"),
    # BASELINE config 4: CelebA-HQ-256 DDPM UNet
    "celebahq256": (dict(model=dict(type="simple", ch=128, out_ch=3, ch_mult=[1, 1, 2, 2, 4, 4], num_res_blocks=2,
                                    attn_resolutions=[16], dropout=0.0, in_channels=3, resamp_with_conv=True, use_fp16=True),
                         diffusion=dict(num_diffusion_timesteps=1000, beta_schedule="linear"),
                         data=dict(dataset="CelebA_HQ", image_size=256, channels=3, subset_1k=False)), "celeba_hq"),
}


def dict2namespace(d):
    ns = argparse.Namespace()
    for k, v in d.items():
        setattr(ns, k, dict2namespace(v) if isinstance(v, dict) else v)
    return ns


def build_parser():
    p = argparse.ArgumentParser()
    a = p.add_argument
    a("--config", type=str, default="cifar10", choices=["cifar10", "imagenet", "celeba", "celeba_hq"])
    a("--config_path", type=str, default="cifar10_adm")
    a("--constraint", type=str, default="none", choices=["none", "sr_bicubic", "sr_averagepooling", "deblur_gauss",
                                                           "colorization", "cs_walshhadamard", "inpainting", "inpainting_half"])
    a("--constraint_proj", type=str, default="svd", choices=["none", "simple", "svd", "simple_gd", "svd_gd", "ddrm"])
    a("--constraint_scale", type=float, default=4.0)
    a("--constraint_lr", type=float, default=10)
    a("--constraint_iter", type=int, default=10)
    a("--constraint_loss", type=str, default="l1", choices=["l1", "l2"])
    a("--prior_xt", type=int, default=0)
    a("--norm_eps", type=int, default=0)
    a("--sigma_type", type=str, default="pred", choices=["base", "pred", "pred_partial"])
    a("--sampling", type=str, default="project", choices=["denoise", "project"])
    a("--norm_init_noise", type=int, default=0)
    a("--redesign_sigma", type=int, default=1)
    a("--min_sigma", type=float, default=0.003)
    a("--max_sigma", type=float, default=0.02)
    a("--sigma_gamma", type=float, default=1.0)
    a("--cycle_size", type=int, default=10)
    a("--max_T", type=int, default=10)
    a("--sampler", type=str, default="ddim_simple_orig", choices=["ddpm", "ddim", "ge", "ddim_simple", "ddim_orig", "ddpm_orig",
                                                                     "ddim_simple_orig", "ddim_simple_drag"])
    a("--num_timesteps", type=int, default=100)
    a("--start_sigma", type=float, default=100)
    a("--end_sigma", type=float, default=0)
    a("--start_t", type=int, default=-1)
    a("--end_t", type=int, default=-1)
    a("--sigma_style", type=str, default="DDIM", choices=["Linear", "DDIM", "Scaled"])
    a("--linear_scale", type=float, default=1.0)
    a("--sampler_var", type=str, default="learned", choices=["learned", "fixedsmall", "fixedlarge", "none"])
    a("--eta", type=float, default=0.85)
    a("--new_eta", type=float, default=None)
    a("--refine_sigma", type=int, default=1)
    a("--continuous_t", type=int, default=1)
    a("--final_alpha_one", type=int, default=1)
    a("--time_shift", type=int, default=0)
    a("--sigma_estimate", type=str, default="1000")
    a("--sigma_pred_threshold", type=int, default=960)
    a("--clip_fn", type=str, default="none", choices=["none", "clamp", "dynamic"])
    a("--recal_sigma_prev", type=int, default=1)
    a("--batch_size", type=int, default=10)
    a("--device", type=str, default="cuda:0")
    a("--precision", type=str, default="", choices=["", "f32", "f32x3", "bf16", "f16"],
      help="HIP-path precision override (not a reference flag).  Default: the model config decides as upstream - use_fp16 -> f16 "
           "operands, else f32.  f32x3 = f32 storage + split-f16 three-pass matrix math (inside 1e-3 of the CPU path, ~3x f32's speed)")
    a("--seed", type=int, default=1234)
    a("--result_dir", type=str, default="results")
    a("--test_dir", type=str, default="temp2")
    a("--sample_size", type=int, default=1000)
    a("--save_folder", type=str, default=None)
    a("--save_flag", type=str, default="0")
    a("--sample_overwrite", type=int, default=0)
    a("--load_folder", type=str, default="7")
    a("--load_eps", type=str, default=None)
    a("--load_sigma", type=str, default="results/cifar10/7/ema_sigma_ckpt_299.pt")
    a("--fid_target", type=str, default=None)
    a("--method", type=str, default="pred_denoise_base",
      choices=["default", "base", "pred_denoise_base", "pred_denoise_proj", "pred_denoise_proj_arbit", "pred_proj",
               "pred_denoise_base_nonorm", "pred_denoise_base_norefine", "pred_partial_denoise_base"])
    # extensions (see module docstring)
    a("--synthetic", type=str, default=None, choices=sorted(SYNTHETIC))
    a("--return_log", type=int, default=0)
    a("--save_png", type=int, default=1)
    return p


def apply_presets(args):
    """Dataset presets and --method presets (image_sample.py:143-268)."""
    per_dataset = {"cifar10": dict(norm_max=54.63, norm_min=0, clip_fn="clamp", sampler_var="learned"),
                   "imagenet": dict(norm_max=440.0, norm_min=0, clip_fn="dynamic", sampler_var="learned"),
                   "celeba": dict(norm_max=110, norm_min=-2, clip_fn="clamp", sampler_var="learned"),
                   "celeba_hq": dict(norm_max=397.0, norm_min=0.0, sampler_var="fixedsmall")}
    args.norm_max = args.norm_min = None
    for k, v in per_dataset.get(args.config, {}).items():
        setattr(args, k, v)
    denoise = dict(sampling="denoise", sigma_style="DDIM", redesign_sigma=0, continuous_t=0)
    table = {
        "base": dict(denoise, sigma_type="base", norm_eps=False, refine_sigma=0),
        "pred_denoise_base": dict(denoise, sigma_type="pred", norm_eps=True, refine_sigma=1),
        "pred_partial_denoise_base": dict(denoise, sigma_type="pred_partial", norm_eps=True, refine_sigma=1),
        "pred_denoise_base_nonorm": dict(denoise, sigma_type="pred", norm_eps=False, refine_sigma=1),
        "pred_denoise_base_norefine": dict(denoise, sigma_type="pred", norm_eps=True, refine_sigma=0),
        "pred_denoise_proj": dict(sampling="denoise", sigma_type="pred", sigma_style="Linear", norm_eps=True, redesign_sigma=0, continuous_t=1),
        "pred_denoise_proj_arbit": dict(sampling="denoise", sigma_type="pred", sigma_style="Linear", norm_eps=True, redesign_sigma=1, continuous_t=1),
    }
    m = args.method
    if m in table:
        for k, v in table[m].items():
            setattr(args, k, v)
        if m == "pred_denoise_proj_arbit" and args.max_T >= 50:
            args.num_timesteps, args.cycle_size = int(0.8 * args.max_T), int(0.1 * args.max_T)
        else:
            args.num_timesteps = args.max_T
    elif "pred_proj" in m:
        for k, v in dict(sampling="project", sigma_type="pred", sigma_style="Linear", norm_eps=True, redesign_sigma=1, continuous_t=1).items():
            setattr(args, k, v)
    if args.sigma_type == "base":
        args.norm_eps, args.sampling, args.redesign_sigma, args.continuous_t, args.refine_sigma = False, "denoise", 0, 0, 0
    else:
        args.norm_eps = True
    return args


def get_args(argv=None):
    args = build_parser().parse_args(argv)
    if args.config_path is None:
        args.config_path = args.config
    rates = [float(x) for x in args.sigma_estimate]
    s = sum(rates)
    rates = [round(x / s, 2) for x in rates]
    rates[0] += 1 - sum(rates)
    args.sigma_estimate_rate = rates
    if args.synthetic:
        cfg, preset = SYNTHETIC[args.synthetic]
        args.config = preset
        config = dict2namespace(cfg)
        saved = dict(load_eps=None, fid_target=None, sigma_block=2, sigma_dropout=0.0, use_sigma_fp16=cfg["model"].get("use_fp16", False))
        args.test_dir = os.path.join(args.test_dir, args.synthetic, args.constraint)
    else:
        import yaml
        args.result_dir = os.path.join(args.result_dir, args.config_path, args.load_folder)
        args.test_dir = os.path.join(args.test_dir, args.config, args.constraint)
        with open(os.path.join(args.result_dir, "args.json")) as f:       # the sigma-net training run's arguments (:112-121)
            saved = json.load(f)
        with open(os.path.join("store", "config", args.config_path + ".yml")) as f:
            config = dict2namespace(yaml.safe_load(f))
    args.load_eps, args.fid_target = saved["load_eps"], saved["fid_target"]
    args.sigma_block = 2 if args.config == "imagenet" else saved["sigma_block"]
    args.sigma_dropout, args.use_sigma_fp16 = saved["sigma_dropout"], saved["use_sigma_fp16"]
    config.model.use_sigma_fp16, config.model.sigma_block, config.model.sigma_dropout = args.use_sigma_fp16, args.sigma_block, args.sigma_dropout
    if "feat_layer" in saved:
        config.model.feat_layer = saved["feat_layer"]
    elif not hasattr(config.model, "feat_layer"):
        config.model.feat_layer = 1
    return apply_presets(args), config


def save_png(img, path):
    from diffusion_nlc_amd.experiments import save_image
    save_image(img, path)


@torch.no_grad()
def evaluate_unconstraint(experiment, n_samples, images_dir, norm_init_noise=False, style="base", sampling="denoise",
                          norm_eps=False, refine_prior_sigma=False, sigma_estimate_rate=(1, 0, 0, 0), max_T=None,
                          sigma_pred_threshold=1000, new_eta=None, recal_sigma_prev=False, return_log=False, save_images=True):
    """image_sample.py:522-569: ceil(n/B) full batches from ONE host generator, skip batches whose PNGs exist.

    Under a launcher (``torchrun --nproc-per-node N image_sample.py ...``, one process per GPU; BASELINE config 5) the batches
    that are to be sampled are dealt round-robin over the ranks.  Every rank walks the WHOLE batch list with the reference's
    single host generator: it samples its own batches and, for a batch another rank owns, draws and discards exactly what that
    batch takes from the generator (initial state + per-step noise of stochastic samplers), so each sample equals the
    single-process run's.  The skip-if-exists decisions (which do not advance the generator upstream either) are taken once, by
    rank 0, before anything is written.  One all-gather (RCCL over xGMI) collects the finished samples; rank 0 writes the PNGs
    and computes FID.  ``return_log`` lists are per rank (rank 0 returns those of its own batches).  Cost to know about: with a
    stochastic sampler (DDPM, eta > 0) every rank draws every batch's per-step noise on the host (world x the randn work per rank);
    deterministic DDIM draws one tensor per batch."""
    from diffusion_nlc_amd import shard
    world, rank = shard.world_rank()
    B = experiment.batch_size
    shape = (B,) + tuple(experiment.data_shape)
    gen = experiment.new_gen()
    n_batches = math.ceil(n_samples / B)
    all_paths = [[os.path.join(images_dir, f"00-{i:05}-{j:03}.png") for j in range(B)] for i in range(n_batches)]
    skip = [bool(save_images and all(os.path.exists(p) for p in paths)) for paths in all_paths]
    skip = shard.broadcast_object(skip)                          # one decision for every rank, taken before any rank writes
    todo = [i for i in range(n_batches) if not skip[i]]
    if world > 1 and sampling == "project" and experiment.host_draws_per_batch(new_eta) != 1:
        raise NotImplementedError("sharded --sampling project needs a deterministic sampler (the projection loop's step count, and so "
                                  "its number of host noise draws, is data dependent)")
    logs, mine = [], []
    for k, i in enumerate(todo):
        if k % world != rank:
            shard.replay_draws(gen, shape, experiment.host_draws_per_batch(new_eta))
            continue
        t1 = time()
        extra = dict(return_on_device=True) if world > 1 else {}
        if sampling == "project":
            sample, return_list = projection_loop(experiment, shape=shape, gen=gen, norm_init_noise=norm_init_noise, style=style,
                                                  constrain_fn=None, norm_eps=norm_eps, refine_prior_sigma=refine_prior_sigma,
                                                  xT=None, return_log=return_log, chunk_size=1,
                                                  sigma_estimate_rate=sigma_estimate_rate, constrain_loss=None, max_T=max_T,
                                                  stop_condition=0.0, sigma_pred_threshold=sigma_pred_threshold, new_eta=new_eta,
                                                  recal_sigma_prev=recal_sigma_prev)
        else:
            sample, return_list = experiment.denoise_loop(shape=shape, gen=gen, norm_init_noise=norm_init_noise, style=style,
                                                          constrain_fn=None, norm_eps=norm_eps,
                                                          refine_prior_sigma=refine_prior_sigma, return_log=return_log,
                                                          chunk_size=1, sigma_pred_threshold=sigma_pred_threshold, new_eta=new_eta,
                                                          **extra)
        print("time:", time() - t1)
        if world > 1 and sampling != "project":
            # a NaN early-break (src/experiments.py:389) leaves draws of a stochastic sampler untaken; the other ranks replayed the
            # full count for this batch, so the owner pads up to it - the ranks stay in step with EACH OTHER (a single-process run,
            # like the reference, carries on from wherever the break left the generator)
            shard.replay_draws(gen, shape, experiment.host_draws_per_batch(new_eta) - getattr(experiment, "host_draws_used", experiment.host_draws_per_batch(new_eta)))
        logs.append(return_list)
        sample = sample.add(1).div(2).clamp(0, 1)
        if world == 1:
            if save_images:
                for img, p in zip(sample, all_paths[i]):
                    save_png(img, p)
        else:
            mine.append(sample)
        print(f"done batches:{i}/{n_batches}")
    if world > 1:
        dev = getattr(experiment, "device", "cpu")
        local = torch.stack(mine) if mine else torch.empty((0,) + shape, device=dev)
        allx = shard.gather_samples(local.to(dev), len(todo), world, rank)        # [len(todo), B, C, H, W] in batch order, every rank
        if rank == 0 and save_images:
            for k, i in enumerate(todo):
                for img, p in zip(allx[k], all_paths[i]):
                    save_png(img, p)
        shard.barrier()                                          # the PNGs are on disk before anyone computes FID / returns
    fid = float("nan")
    if rank == 0 and experiment.fid_fn is not None:
        fid = experiment.fid_fn(images_dir)
    return {"fid": fid}, logs


def projection_loop(self, *args, **kwargs):
    """image_sample.py:431-519 (free function taking the experiment as ``self``): the device implementation is
    ExperimentDiffusion.projection_loop."""
    return self.projection_loop(*args, **kwargs)


def get_constraint_function(args, image_size, channels):
    """image_sample.py:359-405 for the operators on the HIP path: inpainting with the 'svd' (or 'ddrm', same
    operator) projection."""
    from src.constraint_functions import Constraint_Function, svd_constraint
    proj = "svd" if args.constraint_proj == "ddrm" else args.constraint_proj
    if proj != "svd":
        raise NotImplementedError(f"--constraint_proj {args.constraint_proj}: only 'svd' / 'ddrm' are on the HIP path")
    name = args.constraint
    if args.synthetic and name == "inpainting":
        name = "inpainting_random"                 # seeded random 50 % mask instead of store/inp_masks/mask_random.pt
    op = svd_constraint(name, fn_scale=args.constraint_scale, device=args.device, base_mask_dir="store/inp_masks",
                        image_size=image_size, channels=channels)
    return Constraint_Function(args.constraint, op, channels=channels, image_size=image_size, lr=args.constraint_lr)


def synthetic_val_loader(args, shape, n_samples):
    """Stand-in for get_val_loader (image_sample.py:407-428) when no dataset is available: seeded U(0,1) images."""
    g = torch.Generator().manual_seed(args.seed)
    for _ in range(max(1, math.ceil(n_samples / args.batch_size))):
        yield torch.rand((args.batch_size,) + tuple(shape), generator=g), torch.zeros(args.batch_size, dtype=torch.long)


@torch.no_grad()
def evaluate_constraint(experiment, data_loader, Constraint, images_dir, n_samples=-1, norm_init_noise=False, style="base",
                        sampling="denoise", norm_eps=False, refine_prior_sigma=False, prior_xt=False,
                        sigma_estimate_rate=(1, 0, 0, 0), return_log=False, max_T=None, sigma_pred_threshold=1000, new_eta=None,
                        recal_sigma_prev=False, save_images=True):
    """image_sample.py:608-709 (SSIM needs basicsr and is skipped; PSNR / MSE / constraint losses are kept).
    The projection runs fused inside the scheduler kernel (Constraint.bind) unless logging is on."""
    device = experiment.device
    gen = experiment.new_gen()
    mse_list, psnr_list, const_f_loss, const_b_loss, const_orig_loss = [], [], [], [], []
    return_list = None
    for i, (x_orig, _classes) in enumerate(data_loader):
        B = x_orig.shape[0]
        paths = [os.path.join(images_dir, f"00-{i:05}-{j:03}.png") for j in range(B)]
        if save_images and all(os.path.exists(p) for p in paths):
            print("skip images for:", f"00-{i:05}-(000~{B - 1:03}).png")
            continue
        batch_x = 2 * x_orig.to(device) - 1.0
        y = Constraint.transform(batch_x)
        Apy = Constraint.inv_transform(y)
        shape = (B,) + experiment.data_shape
        from functools import partial
        constraint_fn = partial(Constraint.constraint_fn, y=y, lambda_t=Constraint.lr) if return_log else Constraint.bind(y, shape)
        constrain_loss = partial(Constraint.loss, y=y)
        xT = None
        if prior_xt:
            xT = Apy + float(experiment.scheduler.sampling_sigmas[0]) * torch.randn(Apy.shape).to(device)
        t1 = time()
        if sampling == "project":
            sample, return_list = projection_loop(experiment, shape=shape, gen=gen, norm_init_noise=norm_init_noise, style=style,
                                                  constrain_fn=constraint_fn, norm_eps=norm_eps,
                                                  refine_prior_sigma=refine_prior_sigma, xT=xT, return_log=return_log, chunk_size=1,
                                                  sigma_estimate_rate=sigma_estimate_rate, constrain_loss=constrain_loss,
                                                  max_T=max_T, stop_condition=0.0, sigma_pred_threshold=sigma_pred_threshold,
                                                  new_eta=new_eta, recal_sigma_prev=recal_sigma_prev)
        else:
            sample, return_list = experiment.denoise_loop(shape=shape, gen=gen, norm_init_noise=norm_init_noise, style=style,
                                                          constrain_fn=constraint_fn, norm_eps=norm_eps,
                                                          refine_prior_sigma=refine_prior_sigma, xT=xT, return_log=return_log,
                                                          chunk_size=1, constrain_loss=constrain_loss,
                                                          sigma_pred_threshold=sigma_pred_threshold, new_eta=new_eta)
        print("time:", time() - t1)
        sample = sample.add(1).div(2).clamp(0, 1)
        if save_images:
            for img, p in zip(sample, paths):
                save_png(img, p)
        mse = torch.mean((sample - x_orig) ** 2, dim=(1, 2, 3))
        psnr = 10 * torch.log10(1 / mse)
        x_hat = (2 * sample - 1.0).to(device)
        const_f, const_b = Constraint.loss(x_hat, y)
        cons_orig = torch.linalg.vector_norm((x_hat - batch_x).cpu(), ord=1, dim=(1, 2, 3))
        mse_list += mse.tolist(); psnr_list += psnr.tolist()
        const_f_loss += const_f.tolist(); const_b_loss += const_b.tolist(); const_orig_loss += cons_orig.tolist()
        print(f"done batches:{i},  psnr:{np.mean(psnr_list)}, cost:{np.mean(const_f_loss)}")
        if n_samples > 0 and (i + 1) * B > n_samples:
            break
    fid = experiment.fid_fn(images_dir) if experiment.fid_fn is not None else float("nan")
    log_dict = dict(mse=float(np.mean(mse_list)), psner=float(np.mean(psnr_list)), ssim=float("nan"),
                    const_f_loss=float(np.mean(const_f_loss)), const_b_loss=float(np.mean(const_b_loss)),
                    const_orig_loss=float(np.mean(const_orig_loss)), fid=fid,
                    full_log=dict(psnr=psnr_list, mse=mse_list, const_forward=const_f_loss, const_backward=const_b_loss,
                                  const_orig_loss=const_orig_loss))
    return log_dict, return_list


def main(args, config):
    # One process per GPU under a launcher (RANK / WORLD_SIZE / LOCAL_RANK in the environment; INTEGRATION.md shows the torchrun
    # line): joined BEFORE anything touches a device; rank r drives cuda:LOCAL_RANK.  Rank 0 owns the output directory.
    from diffusion_nlc_amd import shard
    rank, world, local = shard.init_from_env()
    if world > 1 and torch.device(args.device).type == "cuda":
        args.device = f"cuda:{local}"
    if args.save_folder is not None:
        args.test_dir = args.save_folder
    else:
        i = 0
        while os.path.exists(os.path.join(args.test_dir, str(i))):
            i += 1
        args.test_dir = shard.broadcast_object(os.path.join(args.test_dir, str(i)))      # every rank uses rank 0's choice
    if rank == 0:
        os.makedirs(args.test_dir, exist_ok=True)
        with open(os.path.join(args.test_dir, "args.json"), "w") as f:
            json.dump({k: (str(v) if k == "device" else v) for k, v in vars(args).items()}, f)
    shard.barrier()
    if args.seed is not None:
        random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)

    mc = config.model
    if mc.type == "openai":
        model, sigma_model, _ = create_sigma_eps_model(**vars(mc))
    else:
        model, sigma_model, _ = create_simple_sigma_eps_model(config)
    print("eps model size:", get_model_size(model))
    print("sigma model size:", get_model_size(sigma_model))
    if args.synthetic:
        from diffusion_nlc_amd.filler import fill_state_dict
        model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
        sigma_model.load_state_dict(fill_state_dict(sigma_model.state_dict(), seed=1, overrides={"final_mlp.weight": 0.1, "final_mlp.bias": 0.5}))
    else:
        model.load_state_dict(torch.load(args.load_eps, map_location="cpu"))
        sigma_model.load_state_dict(torch.load(args.load_sigma, map_location="cpu"))
    if torch.device(args.device).type == "cuda":
        torch.cuda.set_device(torch.device(args.device))       # --device cuda:K: every launch below goes to K's streams
    model.eval().to(args.device)
    sigma_model.eval().to(args.device)
    if getattr(mc, "use_fp16", False):
        model.convert_to_fp16()
    if getattr(mc, "use_sigma_fp16", False):
        sigma_model.convert_to_fp16()
    if getattr(args, "precision", ""):
        dt, mm = {"f32": (torch.float32, "native"), "f32x3": (torch.float32, "f16x3"), "bf16": (torch.bfloat16, "native"),
                  "f16": (torch.float16, "native")}[args.precision]
        for m in (model, sigma_model):
            m.set_compute_dtype(dt).set_matmul(mm)

    dc = config.diffusion
    sampler = get_sampler(args.sampler, dc.num_diffusion_timesteps, args.num_timesteps, beta_schedule=dc.beta_schedule,
                          sigma_style=args.sigma_style, set_alpha_to_one=args.final_alpha_one, start_sigma=args.start_sigma,
                          end_sigma=args.end_sigma, sampler_var=args.sampler_var, continuous_t=args.continuous_t,
                          linear_scale=args.linear_scale, eta=args.eta, norm_eps=args.norm_eps, start_t=args.start_t, end_t=args.end_t)
    if args.redesign_sigma and args.max_T > args.num_timesteps:
        print("redesign sigma", args.num_timesteps, args.max_T)
        redesign_sigma(sampler, args.num_timesteps, args.max_T, args.cycle_size, args.min_sigma, args.max_sigma, args.sigma_gamma)
    sampler.to(args.device)

    d = config.data
    experiment = ImageExperiment(model, sampler, batch_size=args.batch_size, data_shape=(d.channels, d.image_size, d.image_size),
                                 seed=args.seed, device=args.device, save_folder=args.test_dir, dist_train=False,
                                 time_shift=args.time_shift)
    experiment.set_model(model, sigma_model, learn_epsvar=mc.type == "openai")
    experiment.fid_helper(args.fid_target)
    experiment.set_norm_maxmin(args.norm_min, args.norm_max)
    experiment.set_clip_fn(args.clip_fn)

    images_dir = os.path.join(args.test_dir, args.save_flag, "images")
    if rank == 0:
        if os.path.exists(images_dir) and args.sample_overwrite:
            shutil.rmtree(images_dir)
        os.makedirs(images_dir, exist_ok=True)
    shard.barrier()
    if args.constraint != "none" and world > 1:
        raise NotImplementedError("constrained runs select the best x0 by a BATCH-mean loss and read a data loader in order: they are not "
                                  "sharded (SURVEY.md §8e); run them as one process per job")
    if args.constraint == "none":
        log_dict, return_lists = evaluate_unconstraint(
            experiment, args.sample_size, images_dir, norm_init_noise=args.norm_init_noise, style=args.sigma_type,
            sampling=args.sampling, norm_eps=args.norm_eps, refine_prior_sigma=args.refine_sigma,
            sigma_estimate_rate=args.sigma_estimate_rate, max_T=args.max_T, sigma_pred_threshold=args.sigma_pred_threshold,
            new_eta=args.new_eta, recal_sigma_prev=args.recal_sigma_prev, return_log=bool(args.return_log),
            save_images=bool(args.save_png))
        if args.return_log and rank == 0:
            torch.save(return_lists, os.path.join(args.test_dir, args.save_flag, "results_dump.pt"))
    else:
        if not args.synthetic:
            raise NotImplementedError("dataset loaders (datasets/*.py) are out of scope: use --synthetic for constrained runs, or "
                                      "call evaluate_constraint with your own DataLoader")
        Constraint = get_constraint_function(args, d.image_size, d.channels)
        loader = synthetic_val_loader(args, (d.channels, d.image_size, d.image_size), args.sample_size)
        log_dict, _ = evaluate_constraint(
            experiment, loader, Constraint, images_dir, args.sample_size, norm_init_noise=args.norm_init_noise,
            style=args.sigma_type, sampling=args.sampling, norm_eps=args.norm_eps, refine_prior_sigma=args.refine_sigma,
            prior_xt=args.prior_xt, sigma_estimate_rate=args.sigma_estimate_rate, return_log=False, max_T=args.max_T,
            sigma_pred_threshold=args.sigma_pred_threshold, new_eta=args.new_eta, recal_sigma_prev=args.recal_sigma_prev,
            save_images=bool(args.save_png))
    if rank == 0:
        with open(os.path.join(args.test_dir, args.save_flag, "results.json"), "w") as f:
            json.dump(log_dict, f)
    log_dict.pop("full_log", None)
    print(log_dict)
    print("evaluate done")
    if world > 1:
        shard.barrier()
        torch.distributed.destroy_process_group()
    return log_dict


if __name__ == "__main__":
    main(*get_args())
