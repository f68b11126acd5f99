#!/usr/bin/env python3
"""EDM / Heun + NLC sampling entry point with the reference's flags (drop-in for edm_image_sample.py).

Flag names and defaults follow the reference (edm_image_sample.py:19-54); ``main`` keeps its call order
(:110-199): build the SongUNet + sigma net (create_edm_sigma_eps_model), load the EDM weights and the sigma
checkpoint, wrap them in EDMImageExperiment and call evaluate_edm.  Differences on side effects only:
FID / PNG output are optional, and ``--synthetic`` (extension) replaces the NVIDIA pickle download and the
``results/<cfg>/<folder>/args.json`` lookup by a built-in configuration with deterministic filler weights
(the reference ships neither).  With real files, ``--load_eps`` must point to a plain ``state_dict`` (.pt):
unpickling NVIDIA's ``persistence`` classes needs the vendored ``dnnlib/torch_utils`` tree, which is out of scope.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from src.experiments import EDMImageExperiment          # noqa: E402
from src.script_util import create_edm_sigma_eps_model   # noqa: E402
from src.utils import get_model_size                     # noqa: E402

SYNTHETIC = {
    "cifar10": dict(img_resolution=32, in_channels=3, out_channels=3, augment_dim=9, model_channels=128, channel_mult=[2, 2, 2],
                    num_blocks=4, attn_resolutions=[16], dropout=0.13),
    "tiny": dict(img_resolution=32, in_channels=3, out_channels=3, augment_dim=9, model_channels=32, channel_mult=[2, 2, 2],
                 num_blocks=2, attn_resolutions=[16], dropout=0.0),
}


def get_args(argv=None):
    p = argparse.ArgumentParser()
    a = p.add_argument
    a("--config", type=str, default="cifar10", choices=["cifar10", "ffhq", "afhqv2", "imagenet"])
    a("--num_timesteps", type=int, default=18)
    a("--sigma_min", type=float, default=0.002)
    a("--sigma_max", type=float, default=80)
    a("--rho", type=float, default=7)
    a("--S_churn", type=float, default=0)
    a("--S_min", type=float, default=0)
    a("--S_max", type=float, default=float("inf"))
    a("--S_noise", type=float, default=1)
    a("--sigma_type", type=str, default="pred_partial,pred")
    a("--norm_eps", type=str, default="00")
    a("--refine_sigma", type=int, default=0)
    a("--sigma_scheduler", type=str, default="EDM", choices=["EDM", "Linear"])
    a("--eps_ratio", type=float, default=0.5)
    a("--eps_scale", type=float, default=1.0)
    a("--use_second_order", type=int, default=1)
    a("--batch_size", type=int, default=100)
    a("--device", type=str, default="cuda:0")
    a("--seed", type=int, default=0)
    a("--result_dir", type=str, default="results")
    a("--test_dir", type=str, default="temp_edm")
    a("--sample_size", type=int, default=1000)
    a("--save_folder", type=str, default=None)
    a("--save_flag", type=str, default="0")
    a("--load_folder", type=str, default="0")
    a("--load_eps", type=str, default=None)
    a("--load_sigma", type=str, default=None)
    a("--fid_target", type=str, default=None)
    a("--norm_max", type=float, default=54.63)
    a("--norm_min", type=float, default=0.0)
    # extensions
    a("--synthetic", type=str, default=None, choices=sorted(SYNTHETIC))
    a("--dtype", type=str, default="f32", choices=["f32", "f32x3", "bf16", "f16"],
      help="HIP-path precision (not a reference flag): f32 = exact f32 MFMA, f32x3 = f32 storage + split-f16 matrix math, bf16 / f16")
    return p.parse_args(argv)


def main(args):
    # one process per GPU under a launcher: join before anything touches a device, rank r drives cuda:LOCAL_RANK
    from diffusion_nlc_amd import shard
    rank, world, local = shard.init_from_env()
    if world > 1 and torch.device(args.device).type == "cuda":
        args.device = f"cuda:{local}"
    if args.synthetic:
        cfg = dict(SYNTHETIC[args.synthetic])
        saved = dict(sigma_block=2, sigma_dropout=0.0)
    else:
        with open(os.path.join(args.result_dir, args.config, args.load_folder, "args.json")) as f:
            saved = json.load(f)
        cfg = dict(saved["model"]) if "model" in saved else dict(SYNTHETIC["cifar10"])
    cfg.update(sigma_block=saved.get("sigma_block", 2), sigma_dropout=saved.get("sigma_dropout", 0.0))
    model, sigma_model, _ = create_edm_sigma_eps_model(**cfg)
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
        model.load_state_dict(torch.load(args.load_eps, map_location="cpu"))
        sigma_model.load_state_dict(torch.load(args.load_sigma, map_location="cpu"))
    dt, mm = {"f32": (torch.float32, "native"), "f32x3": (torch.float32, "f16x3"), "bf16": (torch.bfloat16, "native"),
              "f16": (torch.float16, "native")}[args.dtype]
    if torch.device(args.device).type == "cuda":
        torch.cuda.set_device(torch.device(args.device))       # --device cuda:K: every launch below goes to K's streams
    model.eval().to(args.device).set_compute_dtype(dt).set_matmul(mm)
    sigma_model.eval().to(args.device).set_compute_dtype(dt).set_matmul(mm)

    res, ch = cfg["img_resolution"], cfg["in_channels"]
    exp = EDMImageExperiment(model, None, batch_size=args.batch_size, data_shape=(ch, res, res), seed=args.seed,
                             device=args.device, save_folder=args.save_folder or args.test_dir, sigma_min=args.sigma_min,
                             sigma_max=args.sigma_max, rho=args.rho, S_churn=args.S_churn, S_min=args.S_min, S_max=args.S_max,
                             S_noise=args.S_noise, num_timesteps=args.num_timesteps)
    exp.set_model(model, sigma_model, learn_epsvar=False)
    exp.fid_helper(args.fid_target)
    exp.set_norm_maxmin(args.norm_min, args.norm_max)
    log_dict, samples = exp.evaluate_edm(args.sample_size, images_dir=None, style=args.sigma_type, norm_eps=args.norm_eps + "0",
                                         refine_prior_sigma=bool(args.refine_sigma), sigma_scheduler=args.sigma_scheduler,
                                         eps_ratio=args.eps_ratio, eps_scale=args.eps_scale,
                                         use_second_order=bool(args.use_second_order))
    print(log_dict, tuple(samples.shape))
    print("evaluate done")
    if world > 1:
        shard.barrier()
        torch.distributed.destroy_process_group()
    return log_dict, samples


if __name__ == "__main__":
    main(get_args())
