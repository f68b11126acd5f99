#!/usr/bin/env python3
"""How precisely does each precision reproduce the NLC-corrected sigma of the first timestep (ADM-256, sigma_0 = 100)?  The first
x0 = xt - sigma * eps turns a relative sigma error d into an x0 error of ~100 d |eps|, so this scalar decides the trajectory parity
of the 16-bit paths.  Oracle: encode + sigma net of B images on the host; HIP: UNet and sigma net precisions varied independently.

    python3 tools/sigma_precision.py [--batch 4]
"""
import argparse
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    args = ap.parse_args()
    B = args.batch
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(tiny=False, batch=B, timesteps=50, dry_run=False, dtype="bf16")
    wl = bench.AdmWorkload(ns, dev, bench.PRECISIONS["bf16"])
    exp = wl.exp
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
    z = torch.randn((B, 3, 256, 256), generator=torch.Generator().manual_seed(4321))
    xT = z / (1 / (s.sampling_sigmas[0] ** 2 + 1)).sqrt()
    with torch.no_grad():
        _, _, st, _ = o.get_denoise_vector(xT, s.timesteps[0], s.sampling_sigmas[0], s.sampling_sigmas[1], "pred", True, True)
    ref = st.reshape(-1).double()
    print("oracle corrected sigma:", [f"{v:.5f}" for v in ref.tolist()])
    S = exp.scheduler
    for unet in ("bf16", "f16", "f32x3"):
        for sig in ("same", "f32x3"):
            if unet == "f32x3" and sig == "f32x3":
                continue
            bench.set_precision(exp.model, bench.PRECISIONS[unet])
            bench.set_precision(exp.sigma_model, bench.PRECISIONS[unet if sig == "same" else sig])
            exp._nlc_step(xT.to(dev), float(S.timesteps[0]), S.sampling_sigmas[0], S.sampling_sigmas[1], "pred", True, True)
            got = exp._state(B)["sigma_t"].cpu().double()
            rel = ((got - ref) / ref)
            print(f"UNet {unet:6s} sigma net {sig:6s}: relative error per image {' '.join(f'{v:+.2e}' for v in rel.tolist())}   max |.| {rel.abs().max():.2e}", flush=True)


if __name__ == "__main__":
    main()
