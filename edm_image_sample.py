#!/usr/bin/env python3
"""EDM / Heun + NLC sampling entry point: drop-in for the reference's edm_image_sample.py.

The command line is the reference's (edm_image_sample.py:19-54): every flag, default and choice list is pinned against the
reference's own parser by ``tests/golden/cli_flags.json`` (tests/test_host_cpu.py::test_cli_flags_match_reference).
``get_args`` keeps its file conventions (:56-107: ``<result_dir>/<config>/<load_folder>/args.json`` of the sigma-net training
run, ``store/config/<config>.yml``, the per-config ``get_default`` presets) and ``main`` its call order (:110-199): output
directory + ``args.json``, seeds, create_edm_sigma_eps_model, load weights, EDMImageExperiment(sigma_min=end_sigma,
sigma_max=start_sigma, sigma_data), fid_helper, set_norm_maxmin, evaluate_edm(sample_size, images_dir, ...), ``results.json``.

Extensions (not reference flags; all default to the reference's behaviour): ``--synthetic`` replaces the two file lookups and
the checkpoints by a built-in configuration with deterministic filler weights (the reference ships neither); ``--dtype`` picks
the HIP-path precision; ``--rho / --S_churn / --S_min / --S_max / --S_noise`` expose EDMImageExperiment's constructor arguments
that the reference leaves at their defaults; ``--save_png 0`` skips the PNG writes.  With real files ``--load_eps`` is a plain
``state_dict`` (.pt) or NVIDIA's network pickle (.pkl, as upstream): the latter is read by ``diffusion_nlc_amd.edm_pickle`` without
the vendored ``dnnlib / torch_utils`` tree and without executing the source text the pickle embeds.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import shutil
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from src.experiments import EDMImageExperiment          # noqa: E402
from src.script_util import create_edm_sigma_eps_model   # noqa: E402
from src.utils import get_model_size                     # noqa: E402

# built-in stand-ins for store/config/<config>.yml (model + data sections) used by --synthetic
SYNTHETIC = {
    "cifar10": dict(model=dict(img_resolution=32, in_channels=3, out_channels=3, augment_dim=9, model_channels=128,
                               channel_mult=[2, 2, 2], num_blocks=4, attn_resolutions=[16], dropout=0.13),
                    data=dict(channels=3, image_size=32)),
    "tiny": dict(model=dict(img_resolution=32, in_channels=3, out_channels=3, augment_dim=9, model_channels=32,
                            channel_mult=[2, 2, 2], num_blocks=2, attn_resolutions=[16], dropout=0.0),
                 data=dict(channels=3, image_size=32)),
}


def dict2namespace(d):
    ns = argparse.Namespace()
    for k, v in d.items():
        setattr(ns, k, dict2namespace(v) if isinstance(v, dict) else v)
    return ns


def build_parser():
    p = argparse.ArgumentParser()
    a = p.add_argument
    a("--config", type=str, default="cifar10", choices=["cifar10", "ffhq"])
    a("--sampler", type=str, default="edm", choices=["edm", "ddim", "euler"])
    a("--sigma_type", type=str, default="pred_partial,pred")
    a("--norm_eps", type=str, default="00")
    a("--num_timesteps", type=int, default=49)
    a("--start_sigma", type=float, default=80)
    a("--end_sigma", type=float, default=0.002)
    a("--sigma_data", type=float, default=0.5)
    a("--sigma_style", type=str, default="EDM", choices=["Linear", "EDM"])
    a("--eps_ratio", type=float, default=0.5)
    a("--eps_scale", type=float, default=1.0)
    a("--eta", type=float, default=1.0)
    a("--refine_sigma", type=int, default=0)
    a("--batch_size", type=int, default=200)
    a("--device", type=str, default="cuda:5")
    a("--seed", type=int, default=1234)
    a("--result_dir", type=str, default="results")
    a("--test_dir", type=str, default="temp")
    a("--sample_size", type=int, default=5000)
    a("--save_folder", type=str, default=None)
    a("--save_flag", type=str, default="0")
    a("--sample_overwrite", type=int, default=0)
    a("--load_folder", type=str, default="6")
    a("--load_eps", type=str, default=None)
    a("--load_sigma", type=str, default="results/ffhq/6/ema_sigma_ckpt_100.pt")
    a("--fid_target", type=str, default=None)
    # extensions (module docstring)
    a("--synthetic", type=str, default=None, choices=sorted(SYNTHETIC))
    a("--dtype", type=str, default="f32", choices=["f32", "f32x3", "bf16", "f16"],
      help="HIP-path precision (not a reference flag): f32 = exact f32 MFMA, f32x3 = f32 storage + split-f16 matrix math, bf16 / f16")
    a("--rho", type=float, default=7)
    a("--S_churn", type=float, default=0)
    a("--S_min", type=float, default=0)
    a("--S_max", type=float, default=float("inf"))
    a("--S_noise", type=float, default=1)
    a("--save_png", type=int, default=1)
    return p


def get_default(args):
    """edm_image_sample.py:86-104: the per-config norm bounds (and ffhq's checkpoint / FID paths)."""
    if args.config == "cifar10":
        args.norm_max = 54.63
        args.norm_min = 0
    elif args.config == "ffhq":
        args.load_eps = "store/models/edm-ffhq-64x64-uncond-vp.pkl"
        args.fid_target = "store/fid/ffhq-64x64.npz"
        args.norm_max = 102.0
        args.norm_min = 0
    else:
        args.norm_max = None
        args.norm_min = None
    return args


def get_args(argv=None):
    args = build_parser().parse_args(argv)
    args.result_dir = os.path.join(args.result_dir, args.config)
    args.root_dir = args.result_dir
    args.result_dir = os.path.join(args.root_dir, args.load_folder)
    args.test_dir = os.path.join(args.test_dir, args.config)
    if args.synthetic:
        saved = dict(load_eps=None, fid_target=None, sigma_block=2, sigma_dropout=0.0, use_sigma_fp16=False, feat_layer=1)
        config = dict2namespace(SYNTHETIC[args.synthetic])
    else:
        import yaml
        with open(os.path.join(args.result_dir, "args.json")) as f:           # the sigma-net training run's arguments (:61-67)
            saved = json.load(f)
        with open(os.path.join("store", "config", args.config + ".yml")) as f:
            config = dict2namespace(yaml.safe_load(f))
    args.load_eps = saved["load_eps"]
    args.fid_target = saved["fid_target"]
    args.sigma_block = saved["sigma_block"]
    args.sigma_dropout = saved["sigma_dropout"]
    args.use_sigma_fp16 = saved["use_sigma_fp16"]
    config.model.use_sigma_fp16 = args.use_sigma_fp16
    config.model.sigma_block = args.sigma_block
    config.model.sigma_dropout = args.sigma_dropout
    config.model.feat_layer = saved["feat_layer"]
    return get_default(args), config


def _load_state(path):
    if str(path).endswith(".pkl"):
        # NVIDIA's network pickle (edm_image_sample.py:152-156 upstream: pickle.load(f)['ema'], then .model.state_dict()), read without
        # dnnlib / torch_utils and without executing the module source it embeds (diffusion_nlc_amd/edm_pickle.py)
        from diffusion_nlc_amd.edm_pickle import load_edm_pickle
        return load_edm_pickle(path)
    return torch.load(path, map_location="cpu")


def main(args, config, return_samples=False):
    # one process per GPU under a launcher: join before anything touches a device, rank r drives cuda:LOCAL_RANK
    from diffusion_nlc_amd import shard
    rank, world, local = shard.init_from_env()
    if world > 1 and torch.device(args.device).type == "cuda":
        args.device = f"cuda:{local}"
    if args.save_folder is not None:                                       # :113-122 (args.json only for a new folder)
        args.test_dir = args.save_folder
        print("save folder", args.save_folder)
        fresh = not os.path.exists(args.test_dir)
    else:                                                                  # :123-135 first free <test_dir>/<i>
        i = 0
        while os.path.exists(os.path.join(args.test_dir, str(i))):
            i += 1
        args.test_dir = shard.broadcast_object(os.path.join(args.test_dir, str(i)))      # every rank uses rank 0's choice
        fresh = True
    if rank == 0 and fresh:
        os.makedirs(args.test_dir, exist_ok=True)
        with open(os.path.join(args.test_dir, "args.json"), "w") as f:
            json.dump({k: (str(v) if k == "device" else v) for k, v in vars(args).items()}, f)
    shard.barrier()
    print("args:", args)
    print("config:", config)
    if args.seed is not None:
        random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)

    model, sigma_model, _ = create_edm_sigma_eps_model(**vars(config.model))
    print("eps model size:", get_model_size(model))
    print("sigma model size:", get_model_size(sigma_model))
    if args.synthetic:
        from diffusion_nlc_amd.filler import fill_state_dict
        tmpl = model.state_dict()
        for k in tmpl:                                  # architecture constants: f.ger(f)/f.sum()^2 with f = [1,1]
            if k.endswith("resample_filter"):
                tmpl[k] = torch.ones_like(tmpl[k]) / 4.0
        model.load_state_dict(fill_state_dict(tmpl, seed=0))
        sigma_model.load_state_dict(fill_state_dict(sigma_model.state_dict(), seed=1,
                                                    overrides={"final_mlp.weight": 0.1, "final_mlp.bias": 0.5}))
    else:
        model.load_state_dict(_load_state(args.load_eps))
        print("load eps model from", args.load_eps)
        sigma_model.load_state_dict(_load_state(args.load_sigma))
        print("load sigma model from", args.load_sigma)
    dt, mm = {"f32": (torch.float32, "native"), "f32x3": (torch.float32, "f16x3"), "bf16": (torch.bfloat16, "native"),
              "f16": (torch.float16, "native")}[args.dtype]
    if torch.device(args.device).type == "cuda":
        torch.cuda.set_device(torch.device(args.device))       # --device cuda:K: every launch below goes to K's streams
    model.eval().to(args.device).set_compute_dtype(dt).set_matmul(mm)
    sigma_model.eval().to(args.device).set_compute_dtype(dt).set_matmul(mm)

    d = config.data
    experiment = EDMImageExperiment(model, scheduler=None, batch_size=args.batch_size,
                                    data_shape=(d.channels, d.image_size, d.image_size), seed=args.seed, device=args.device,
                                    save_folder=args.test_dir, dist_train=False, num_timesteps=args.num_timesteps,
                                    sigma_min=args.end_sigma, sigma_max=args.start_sigma, sigma_data=args.sigma_data,
                                    rho=args.rho, S_churn=args.S_churn, S_min=args.S_min, S_max=args.S_max, S_noise=args.S_noise)
    experiment.set_model(model, sigma_model, learn_epsvar=False)
    experiment.fid_helper(args.fid_target)
    experiment.set_norm_maxmin(args.norm_min, args.norm_max)

    images_dir = os.path.join(args.test_dir, args.save_flag, "images")
    if rank == 0:
        if os.path.exists(images_dir) and args.sample_overwrite:
            shutil.rmtree(images_dir)
        os.makedirs(images_dir, exist_ok=True)
    shard.barrier()
    gen = experiment.new_gen()
    log_dict = experiment.evaluate_edm(args.sample_size, images_dir, gen=gen, style=args.sigma_type, norm_eps=args.norm_eps,
                                       refine_prior_sigma=args.refine_sigma, sigma_scheduler=args.sigma_style,
                                       eps_ratio=args.eps_ratio, eps_scale=args.eps_scale,
                                       use_second_order=args.sampler == "edm", save_images=bool(args.save_png))
    if rank == 0:
        with open(os.path.join(args.test_dir, args.save_flag, "results.json"), "w") as f:
            json.dump(log_dict, f)
    print(log_dict)
    print("evaluate done")
    if world > 1:
        shard.barrier()
        torch.distributed.destroy_process_group()
    return (log_dict, experiment.last_samples) if return_samples else log_dict


if __name__ == "__main__":
    main(*get_args())
