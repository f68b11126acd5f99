#!/usr/bin/env python3
"""Per-timestep parity of the HIP path against the CPU oracle on the headline model (ADM-256, DDIM+NLC, B = 1), in every
precision bench.py can run: the oracle steps N timesteps once (~1.4 s each on the GPU box's 16 host cores; N = 50 is the whole
sample), every precision then runs the same seeded x_T through exp.denoise_loop with logging, and the clipped x0 estimate and
the NLC-corrected sigma of every timestep are compared.  Writes one JSON (default gpurun_out/parity_trace.json).

    python3 tools/parity_trace.py --timesteps 50 --precisions bf16 f16 f32x3 f32
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--timesteps", type=int, default=50)
    ap.add_argument("--precisions", nargs="+", default=["bf16", "f16", "f32x3", "f32"])
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "parity_trace.json"))
    ap.add_argument("--conditioning", action="store_true",
                    help="also run the ORACLE itself a second time from x_T perturbed by one f32 ulp per element (random sign): how far two "
                         "runs of the reference arithmetic drift apart over the same timesteps - the floor for ANY independent implementation")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(tiny=False, batch=1, timesteps=50, dry_run=False, dtype="f32")
    wl = bench.AdmWorkload(ns, dev, bench.PRECISIONS["f32"])
    # the oracle leg of bench.py, unbounded: N timesteps of the real trajectory
    from diffusion_nlc_amd.filler import fill_state_dict
    from diffusion_nlc_amd.script_util import create_sigma_eps_model
    from oracle import adm
    from oracle.loop import DiffusionOracle
    from oracle.sched import get_sampler
    cfg = wl.cfg
    torch.set_num_threads(bench._host_cores())
    ucfg, scfg, _ = adm.configs_from_factory(**cfg)
    eps_m, sig_m, _ = create_sigma_eps_model(**cfg)
    sd_e = fill_state_dict(eps_m.state_dict(), seed=0)
    sd_s = fill_state_dict(sig_m.state_dict(), seed=1, overrides=bench.SIGMA_OVERRIDES)
    s = get_sampler("ddim", 1000, 50, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="learned", eta=0.0)
    o = DiffusionOracle(lambda x, t: adm.unet(sd_e, ucfg, x, t, "forward"), lambda x, t: adm.unet(sd_e, ucfg, x, t, "encode"),
                        lambda f: adm.sigma_net(sd_s, scfg, f), s, (3, 256, 256), learn_epsvar=True, norm_min=0.0, norm_max=440.0,
                        clip_fn="dynamic")
    z = torch.randn((1, 3, 256, 256), generator=torch.Generator().manual_seed(1234))
    xT = z / (1 / (s.sampling_sigmas[0] ** 2 + 1)).sqrt()
    def oracle_run(x_start, tag):
        xt, x0s, sigs = x_start, [], []
        t0 = time.perf_counter()
        with torch.no_grad():
            for n in range(args.timesteps):
                eps, lv, st, sp = o.get_denoise_vector(xt, s.timesteps[n], s.sampling_sigmas[n], s.sampling_sigmas[n + 1], "pred", True, True)
                x0 = o.clip(s.pred_xstart(xt, eps, st))
                x0s.append(x0.clone()); sigs.append(st.reshape(-1).clone())
                xt = s.pred_xprev(x0=x0, eps=eps, sigma_t=st, sigma_prev=sp, xt=xt, log_variance=lv)
                if n % 5 == 4:
                    print(f"{tag}: {n + 1} timesteps, {time.perf_counter() - t0:.0f} s", flush=True)
        return x0s, sigs, time.perf_counter() - t0

    x0_ref, sig_ref, secs = oracle_run(xT, "oracle")
    out = {"model": "ADM-256 (filler weights), DDIM+NLC 50-step schedule, B=1, seed 1234", "timesteps": args.timesteps,
           "oracle_seconds": secs, "x0_rms_final": float(x0_ref[-1].double().pow(2).mean().sqrt()), "precisions": {}}
    if args.conditioning:
        sign = torch.randint(0, 2, xT.shape, generator=torch.Generator().manual_seed(1)).float() * 2 - 1
        xT2 = (xT * (1 + sign * 2.0 ** -23)).float()              # one ulp per element, random sign
        x0_b, sig_b, _ = oracle_run(xT2, "oracle (x_T perturbed by 1 ulp)")
        linf = [float((x0_b[i].double() - x0_ref[i].double()).abs().max()) for i in range(args.timesteps)]
        rms = [float((x0_b[i].double() - x0_ref[i].double()).pow(2).mean().sqrt()) for i in range(args.timesteps)]
        out["oracle_vs_oracle_1ulp"] = {"x0_linf_per_timestep": linf, "x0_rms_per_timestep": rms, "final_linf": linf[-1], "max_linf": max(linf),
                                        "what": "the CPU oracle (the reference's arithmetic) against itself, x_T perturbed by one f32 ulp per element"}
        print(f"oracle vs oracle (1 ulp): x0 L-inf first {linf[0]:.3e}  max {max(linf):.3e}  final {linf[-1]:.3e}  RMS final {rms[-1]:.3e}", flush=True)
    for name in args.precisions:
        for m in (wl.exp.model, wl.exp.sigma_model):
            bench.set_precision(m, bench.PRECISIONS[name])
        _, logs = wl.exp.denoise_loop(shape=(1, 3, 256, 256), xT=xT, style="pred", norm_eps=True, refine_prior_sigma=True, return_log=True,
                                      chunk_size=1, sigma_pred_threshold=960, max_steps=args.timesteps)
        linf = [float((logs[3][i].double() - x0_ref[i].double()).abs().max()) for i in range(args.timesteps)]
        rms = [float((logs[3][i].double() - x0_ref[i].double()).pow(2).mean().sqrt()) for i in range(args.timesteps)]
        srel = [float(((wl.exp.sigma_trace[i].double() - sig_ref[i].double()).abs() / sig_ref[i].double()).max()) for i in range(args.timesteps)]
        out["precisions"][name] = {"x0_linf_per_timestep": linf, "x0_rms_per_timestep": rms, "sigma_rel_per_timestep": srel,
                                   "final_linf": linf[-1], "final_rms": rms[-1], "max_linf": max(linf)}
        print(f"{name}: x0 L-inf first {linf[0]:.3e}  max {max(linf):.3e}  final {linf[-1]:.3e}   RMS final {rms[-1]:.3e}   "
              f"sigma rel first {srel[0]:.3e} max {max(srel):.3e}", flush=True)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(out))
    print("wrote", args.out)


if __name__ == "__main__":
    main()
