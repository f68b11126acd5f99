#!/usr/bin/env python3
"""Headline benchmark: images/sec, 256x256 50-step DDIM + NLC on the ADM-256 UNet (BASELINE.json
configs[1]: batch 16 per GPU, bf16 operands / f32 accumulate, synthetic inputs and filler weights).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full 50-timestep DDIM+NLC sampling pass of one batch (16 images) per GPU:
per timestep  UNet.encode -> sigma net -> UNet.forward -> scheduler update.  Every rank samples its
own batches (weak scaling, independent samples); the single collective is one RCCL
all-gather of the finished samples inside the timed region.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     the dominant kernel (conv_igemm_kernel, bf16 MFMA): algorithmic direct-conv FLOPs of its
               launches / their summed duration, measured with HIP events on the launch stream during
               the timed region; peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md).
  cpu_baseline the CPU oracle (oracle/, a port of the reference's PyTorch-CPU path) timed on the host
               cores of this box on ONE ADM-256 DDIM+NLC timestep at B=1, extrapolated x50.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ADM256 = dict(image_size=256, num_channels=256, num_res_blocks=2, channel_mult="", learn_sigma=True,
              attention_resolutions="32,16,8", num_heads=4, num_head_channels=64, use_scale_shift_norm=True,
              resblock_updown=True, use_new_attention_order=False, sigma_block=2)
# per image, one DDIM+NLC timestep: forward + encode + sigma net (BASELINE.md §2, 2*MAC of conv/linear/bmm)
GF_PER_IMAGE_STEP = 2239.67 + 580.29 + 3.95
PEAK_BF16_DENSE_TFLOPS = 2500.0
SIGMA_OVERRIDES = {"final_mlp.weight": 0.1, "final_mlp.bias": 0.5}


def build_models(cfg, device, dtype):
    from diffusion_nlc_amd import script_util
    from diffusion_nlc_amd.filler import fill_state_dict
    eps, sig, fshape = script_util.create_sigma_eps_model(**cfg)
    eps.load_state_dict(fill_state_dict(eps.state_dict(), seed=0))
    sig.load_state_dict(fill_state_dict(sig.state_dict(), seed=1, overrides=SIGMA_OVERRIDES))
    eps.to(device).set_compute_dtype(dtype)
    sig.to(device).set_compute_dtype(dtype)
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


def cpu_baseline(cfg, max_seconds=240.0):
    """Time the oracle (CPU port) on one DDIM+NLC timestep of ADM-256 at B=1."""
    from diffusion_nlc_amd.filler import fill_state_dict
    from diffusion_nlc_amd.script_util import create_sigma_eps_model
    from oracle import adm
    from oracle.loop import DiffusionOracle
    from oracle.sched import get_sampler
    # the GPU box shares its host: a 1-GPU slot owns 16 cores (os.cpu_count() reports the whole machine)
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
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
        return s.pred_xprev(x0=x0, eps=eps, sigma_t=st, sigma_prev=sp, xt=x, log_variance=lv)

    # bounded sample: keep stepping the real trajectory until ~15 s of CPU work (at most 20 of the 50 timesteps)
    n, t0 = 0, time.perf_counter()
    with torch.no_grad():
        while n < 20 and (n == 0 or time.perf_counter() - t0 < 15.0):
            xt = one_timestep(xt, n)
            n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": 1.0 / (50.0 * dt), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} DDIM+NLC timestep(s) of ADM-{res} at B=1 in f32 on the host ({dt:.2f} s each), extrapolated to 50 timesteps"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--timesteps", type=int, default=50, help="DDIM timesteps per sample (50 = the headline metric)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="64x64 debugging configuration (NOT the headline metric)")
    args = ap.parse_args()

    from diffusion_nlc_amd import ops, shard
    rank, world, local = shard.init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cfg = dict(ADM256)
    if args.tiny:
        cfg.update(image_size=64, num_channels=64, channel_mult="1,2,2,4", attention_resolutions="16,8", num_head_channels=32)
    res = cfg["image_size"]
    exp = make_experiment(cfg, device, dtype, args.batch, args.timesteps)
    shape = (args.batch, 3, res, res)
    n_total = (args.warmup + args.steps) * world
    zs = shard.draw_initial_noise(shape, n_total, 1234, world, rank)          # host, reference draw order
    sigma0 = exp.scheduler.sampling_sigmas[0]
    xTs = [(z / (1 / (sigma0 ** 2 + 1)).sqrt()).to(device) for z in zs]       # resident in HBM before timing

    def one(xT):
        x, _ = exp.denoise_loop(shape=shape, xT=xT, style="pred", norm_eps=True, refine_prior_sigma=True,
                                return_log=False, chunk_size=1, sigma_pred_threshold=960)
        return x

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one(xTs[i])
    prof = None if args.no_roofline else []
    barrier()
    ops.CONV_PROFILE = prof
    t0 = time.perf_counter()
    outs = []
    for i in range(args.steps):
        outs.append(one(xTs[args.warmup + i]).to(device))
    local_out = torch.stack(outs)
    gathered = shard.gather_samples(local_out, args.steps * world, world, rank)     # the one RCCL all-gather
    barrier()
    elapsed = time.perf_counter() - t0
    ops.CONV_PROFILE = None
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
    assert gathered.shape[0] == args.steps * world and torch.isfinite(gathered).all()

    images = args.batch * args.steps * world
    line = {
        "metric": "images/sec whole-node, 256x256 50-step DDIM+NLC" if (res == 256 and args.timesteps == 50)
                  else f"images/sec, {res}x{res} {args.timesteps}-step DDIM+NLC (debug configuration)",
        "value": images / elapsed, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"ADM UNet {res}x{res} (src/unet_adm.py), {args.timesteps}-step DDIM+NLC, batch {args.batch} per GPU, "
                               f"{args.dtype}, dynamic-threshold clip, learned variance, filler weights",
                   "global_batch": args.batch * world, "parallelism": f"dp{world} (independent samples, one all-gather)"},
    }
    if rank == 0:
        if prof:
            tot_ms = sum(e0.elapsed_time(e1) for e0, e1, _, d, _s in prof if d == dtype)
            tot_fl = sum(f for _, _, f, d, _s in prof if d == dtype)
            n = sum(1 for _, _, _, d, _s in prof if d == dtype)
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            peak = PEAK_BF16_DENSE_TFLOPS if dtype == torch.bfloat16 else 157.3
            line["roofline"] = {"bound": "mfma", "kernel": f"nlc_conv2d: conv_halo_kernel<{args.dtype}> (3x3 on >= 32x32 maps, 75 % of its time) + conv_fast_kernel<{args.dtype},9|1> (+ splitk_reduce) + conv_igemm_kernel (strided)",
                                "achieved": ach, "peak": peak,
                                "unit": "TFLOP/s", "frac": ach / peak, "traffic": None, "launches": n,
                                "avg_launch_us": 1e3 * tot_ms / max(n, 1), "avg_launch_gflop": tot_fl / max(n, 1) / 1e9,
                                "share_of_wall": tot_ms * 1e-3 / elapsed}
            # HBM bytes per launch of the dominant conv shape, from the separate rocprofv3 --pmc passes of the same build
            # (profiles/r01c_pmc_traffic.json: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE); that shape's own
            # launch time comes from this run's HIP events.
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01c_pmc_traffic.json")
            if res == 256 and dtype == torch.bfloat16 and os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f)
                dom = [(e0.elapsed_time(e1), fl) for e0, e1, fl, d, shp in prof
                       if d == dtype and shp[:4] == (args.batch * res * res, 256, 9, 256) and shp[5] == 0]
                dominant = {"kernel": tj["kernel"], "traffic": tj["traffic_bytes_per_launch"],
                            "algorithmic_bytes": tj["algorithmic_bytes_per_launch"],
                            "launch_us": 1e3 * sum(t for t, _ in dom) / max(len(dom), 1),
                            "tflops": sum(fl for _, fl in dom) / max(sum(t for t, _ in dom), 1e-9) / 1e9}
                # per-launch average over ALL nlc_conv2d launches of a bench step (same population as `achieved`), from
                # PMC passes over two NLC timesteps of this workload: profiles/r01f_pmc_traffic_all.json
                apath = os.path.join(os.path.dirname(tpath), "r01f_pmc_traffic_all.json")
                if os.path.exists(apath):
                    with open(apath) as f:
                        line["roofline"]["traffic"] = json.load(f)["conv2d_bytes_per_launch"]
                else:
                    line["roofline"]["traffic"] = tj["traffic_bytes_per_launch"]
                line["roofline"]["traffic_of_dominant_launch"] = dominant
            if os.environ.get("NLC_BENCH_SHAPES"):
                agg = {}
                for e0, e1, f, d, shp in prof:
                    a = agg.setdefault((str(d), shp), [0, 0.0, 0.0])
                    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += f
                for (d, shp), (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
                    print(f"# conv {d} M={shp[0]} N={shp[1]} taps={shp[2]} Cin={shp[3]} s={shp[4]} ups={shp[5]} C1={shp[6]}: "
                          f"{cnt} launches, {ms:.1f} ms, {fl / ms / 1e9:.0f} TFLOP/s", file=sys.stderr)
            if res == 256:
                line["end_to_end_tflops_per_gpu"] = (images / world) * args.timesteps * GF_PER_IMAGE_STEP / 1e3 / elapsed
        if not args.no_cpu_baseline and not args.tiny:
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
