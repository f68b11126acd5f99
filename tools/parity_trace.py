#!/usr/bin/env python3
"""Per-timestep parity of the HIP path against the CPU oracle on the headline model (ADM-256, DDIM+NLC, B = 1), in every
precision bench.py can run: the oracle steps N timesteps once (~1.4 s each on the GPU box's 16 host cores; N = 50 is the whole
sample), every precision then runs the same seeded x_T through exp.denoise_loop with logging, and the clipped x0 estimate and
the NLC-corrected sigma of every timestep are compared.  Writes one JSON (default gpurun_out/parity_trace.json).

    python3 tools/parity_trace.py --timesteps 50 --precisions bf16 f16 f32x3 f32

The third leg (--make-f64 / --f64-trace): the SAME oracle graph evaluated in float64 end to end (state_dict, schedule tables and
x_T cast to double; oracle/adm.py keeps float64 tensors in float64) - the function that the reference's f32 arithmetic and the HIP
path both approximate.  `--make-f64 FILE` needs no GPU (≈ 30 s per timestep on 8 cores): it writes the per-timestep clipped x0 and
corrected sigma of the f64 run to FILE; `--f64-trace FILE` on the GPU box then reports, per timestep, |CPU-f32 - f64| next to
|HIP - f64| for every precision: an independent f32 implementation is as good as the reference when its distance to the f64
trajectory is no larger than the reference arithmetic's own.
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
    ap.add_argument("--make-f64", default="", metavar="FILE", help="CPU only: run the oracle in float64 and save its trajectory (no GPU needed)")
    ap.add_argument("--f64-trace", default="", metavar="FILE", help="a trajectory written by --make-f64: adds the distances to it")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--mixed", nargs="*", type=int, default=[], metavar="K",
                    help="sigma-adaptive precision: timesteps 0..K-1 in f32x3, K..end in --mixed-low, for every K given (the error a timestep "
                         "injects into the state is (sigma_t - sigma_prev) * d(eps), which shrinks with sigma)")
    ap.add_argument("--mixed-low", default="f16", choices=["f16", "bf16"])
    args = ap.parse_args()
    ns = argparse.Namespace(tiny=False, batch=1, timesteps=50, dry_run=bool(args.make_f64), dtype="f32")
    dev = torch.device("cpu" if args.make_f64 else "cuda:0")
    wl = bench.AdmWorkload(ns, dev, bench.PRECISIONS["f32"])
    # the oracle leg of bench.py, unbounded: N timesteps of the real trajectory
    from diffusion_nlc_amd.filler import fill_state_dict
    from diffusion_nlc_amd.script_util import create_sigma_eps_model
    from oracle import adm
    from oracle.loop import DiffusionOracle
    from oracle.sched import get_sampler
    cfg = wl.cfg
    torch.set_num_threads(args.threads or bench._host_cores())
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
    o32, s32 = o, s

    def oracle_run(x_start, tag, o=None, s=None):
        o, s = o or o32, s or s32
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

    if args.make_f64:
        # float64 leg: same parameters (the f32 state_dict and the f32 schedule tables, cast), double arithmetic throughout;
        # the discrete sigma -> t lookup stays a searchsorted on the (cast) table
        sd_e64 = {k: v.double() for k, v in sd_e.items()}
        sd_s64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd_s.items()}
        s64 = get_sampler("ddim", 1000, 50, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="learned", eta=0.0)
        for k, v in list(vars(s64).items()):
            if torch.is_tensor(v) and v.dtype == torch.float32:
                setattr(s64, k, v.double())
        assert torch.equal(s64.sampling_sigmas.float(), s32.sampling_sigmas) and torch.equal(s64.timesteps, s32.timesteps)
        o64 = DiffusionOracle(lambda x, t: adm.unet(sd_e64, ucfg, x, t, "forward"), lambda x, t: adm.unet(sd_e64, ucfg, x, t, "encode"),
                              lambda f: adm.sigma_net(sd_s64, scfg, f), s64, (3, 256, 256), learn_epsvar=True, norm_min=0.0,
                              norm_max=440.0, clip_fn="dynamic")
        x0_64, sig_64, secs = oracle_run(xT.double(), "oracle f64", o64, s64)
        assert x0_64[0].dtype == torch.float64 and sig_64[0].dtype == torch.float64
        Path(args.make_f64).parent.mkdir(parents=True, exist_ok=True)
        torch.save({"x0": torch.stack(x0_64), "sigma": torch.stack(sig_64), "xT": xT, "timesteps": args.timesteps, "seconds": secs,
                    "torch": torch.__version__, "threads": torch.get_num_threads()}, args.make_f64)
        print("wrote", args.make_f64, f"({secs:.0f} s)")
        return
    x0_ref, sig_ref, secs = oracle_run(xT, "oracle")
    out = {"model": "ADM-256 (filler weights), DDIM+NLC 50-step schedule, B=1, seed 1234", "timesteps": args.timesteps,
           "oracle_seconds": secs, "x0_rms_final": float(x0_ref[-1].double().pow(2).mean().sqrt()), "precisions": {}}
    f64 = None
    if args.f64_trace:
        f64 = torch.load(args.f64_trace, weights_only=False)          # our own file (written by --make-f64)
        assert torch.equal(f64["xT"], xT) and f64["timesteps"] >= args.timesteps

        def to_f64(x0s, sigs):
            n = len(x0s)
            linf = [float((x0s[i].double().cpu() - f64["x0"][i]).abs().max()) for i in range(n)]
            rms = [float((x0s[i].double().cpu() - f64["x0"][i]).pow(2).mean().sqrt()) for i in range(n)]
            srel = [float(((sigs[i].double().cpu().view(-1) - f64["sigma"][i].view(-1)).abs() / f64["sigma"][i].view(-1)).max()) for i in range(n)]
            return {"x0_linf_per_timestep": linf, "x0_rms_per_timestep": rms, "sigma_rel_per_timestep": srel}

        out["f64"] = {"what": "distances to the float64 evaluation of the same graph (oracle/adm.py in double, same parameters, "
                              f"torch {f64['torch']}, {f64['seconds']:.0f} s on {f64['threads']} threads); cpu_f32 = the oracle = the reference's arithmetic",
                      "cpu_f32": to_f64(x0_ref, sig_ref)}
        c = out["f64"]["cpu_f32"]
        print(f"CPU-f32 oracle vs f64: x0 L-inf first {c['x0_linf_per_timestep'][0]:.3e}  t10 {c['x0_linf_per_timestep'][min(9, args.timesteps - 1)]:.3e}  "
              f"final {c['x0_linf_per_timestep'][-1]:.3e}  RMS final {c['x0_rms_per_timestep'][-1]:.3e}", flush=True)
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
        if f64 is not None:
            d = to_f64([logs[3][i] for i in range(args.timesteps)], [wl.exp.sigma_trace[i] for i in range(args.timesteps)])
            c = out["f64"]["cpu_f32"]
            d["ratio_to_cpu_f32_linf"] = [a / max(b, 1e-30) for a, b in zip(d["x0_linf_per_timestep"], c["x0_linf_per_timestep"])]
            d["ratio_to_cpu_f32_rms"] = [a / max(b, 1e-30) for a, b in zip(d["x0_rms_per_timestep"], c["x0_rms_per_timestep"])]
            out["f64"][name] = d
            print(f"{name} vs f64: x0 L-inf first {d['x0_linf_per_timestep'][0]:.3e}  final {d['x0_linf_per_timestep'][-1]:.3e}  RMS final "
                  f"{d['x0_rms_per_timestep'][-1]:.3e};  ratio to CPU-f32's own distance: L-inf max {max(d['ratio_to_cpu_f32_linf']):.2f} "
                  f"median {sorted(d['ratio_to_cpu_f32_linf'])[len(d['ratio_to_cpu_f32_linf']) // 2]:.2f}, RMS max {max(d['ratio_to_cpu_f32_rms']):.2f}", flush=True)
    if args.mixed:
        N = args.timesteps
        run = lambda **kw: wl.exp.denoise_loop(shape=(1, 3, 256, 256), style="pred", norm_eps=True, refine_prior_sigma=True, return_log=True,
                                               chunk_size=1, sigma_pred_threshold=960, max_steps=N, **kw)
        for m in (wl.exp.model, wl.exp.sigma_model):
            bench.set_precision(m, bench.PRECISIONS["f32x3"])
        x_hi, _ = run(xT=xT)
        xts = [t.clone() for t in wl.exp.xt_trace]
        out["mixed"] = {"low": args.mixed_low, "what": f"timesteps 0..K-1 in f32x3, K..{N - 1} in {args.mixed_low}; final clipped x0 vs the CPU-f32 oracle"
                                                       + (" and vs the f64 trajectory" if f64 is not None else ""), "K": {}}
        for K in sorted(set(args.mixed)):
            if not 0 <= K <= N:
                continue
            if K == N:
                x = x_hi
            else:
                for m in (wl.exp.model, wl.exp.sigma_model):
                    bench.set_precision(m, bench.PRECISIONS[args.mixed_low])
                x, _ = run(xT=xts[K], start_step=K)
            d = x.double().cpu() - x0_ref[N - 1].double()
            row = {"final_linf_vs_oracle": float(d.abs().max()), "final_rms_vs_oracle": float(d.pow(2).mean().sqrt())}
            if f64 is not None:
                d64 = x.double().cpu() - f64["x0"][N - 1]
                row.update(final_linf_vs_f64=float(d64.abs().max()), final_rms_vs_f64=float(d64.pow(2).mean().sqrt()))
            out["mixed"]["K"][str(K)] = row
            print(f"mixed f32x3[0,{K}) + {args.mixed_low}[{K},{N}): " + "  ".join(f"{k} {v:.3e}" for k, v in row.items()), flush=True)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(out))
    print("wrote", args.out)


if __name__ == "__main__":
    main()
