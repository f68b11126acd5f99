"""Pin the CPU oracle (oracle/) against the golden vectors captured from the reference itself.

Runs on CPU (no GPU, no /root/reference).  The oracle calls the same ATen CPU kernels in the
same order as the reference, so agreement is at the level of a few f32 ulps; tolerances are
1e-5 absolute on O(1) quantities (1e-4 for the multi-step loops, where per-step rounding
differences in reductions are amplified by the network).
"""
import json

import pytest
import torch

from tests.util import load_npz, load_specs, max_err, oracle_nets, state_dicts

torch.set_num_threads(max(1, min(8, torch.get_num_threads())))


def test_filler_checksums_match_reference_modules():
    from diffusion_nlc_amd.filler import checksum
    specs = load_specs()
    for tag in ("adm_tiny", "adm_tiny_b", "simple_tiny", "edm_tiny"):
        e, s = state_dicts(tag)
        for got, ref in ((checksum(e), specs[tag]["eps_checksum"]), (checksum(s), specs[tag]["sigma_checksum"])):
            assert abs(got[0] - ref[0]) < 1e-6 * max(1, abs(ref[0])) and abs(got[1] - ref[1]) < 1e-6 * abs(ref[1]), tag


@pytest.mark.parametrize("tag", ["adm_tiny", "adm_tiny_b", "simple_tiny", "edm_tiny"])
def test_nets_match_reference(tag):
    g = load_npz(f"net_{tag}")
    eps_fn, enc_fn, sig_fn, both_fn = oracle_nets(tag)
    with torch.no_grad():
        out = eps_fn(g["x"], g["t"])
        feat = enc_fn(g["x"], g["t"])
        r = sig_fn(feat)
        assert max_err(out, g["out"]) < 1e-5, max_err(out, g["out"])
        assert max_err(feat, g["feat"]) < 1e-5
        assert max_err(r, g["r"]) < 1e-5
        assert out.abs().max() > 1e-2 and r.abs().max() > 1e-3      # non-trivial signal (no zero-init collapse)
        if both_fn is not None:
            o2, f2 = both_fn(g["x"], g["t"])
            assert torch.equal(o2, out) and torch.equal(f2, feat)


def test_class_conditional_adm_matches_reference():
    """model(x, t, y): the label-embedding row is added to the timestep embedding (src/unet_adm.py:479-480,652-654)."""
    from oracle import adm
    from tests.util import load_specs
    g = load_npz("net_adm_tiny_cc")
    sd_e, sd_s = state_dicts("adm_tiny_cc")
    ucfg, scfg, _ = adm.configs_from_factory(**load_specs()["_configs"]["adm_tiny_cc"])
    with torch.no_grad():
        out = adm.unet(sd_e, ucfg, g["x"], g["t"], "forward", y=g["y"])
        feat = adm.unet(sd_e, ucfg, g["x"], g["t"], "encode", y=g["y"])
        assert max_err(out, g["out"]) < 1e-5 and max_err(feat, g["feat"]) < 1e-5
        assert max_err(adm.sigma_net(sd_s, scfg, feat), g["r"]) < 1e-5
        assert max_err(adm.unet(sd_e, ucfg, g["x"], g["t"], "forward", y=torch.zeros_like(g["y"])), g["out"]) > 1e-4


def test_scheduler_tables_and_lookups():
    from oracle.sched import get_sampler
    g = load_npz("sched")
    for steps in (10, 50, 100):
        s = get_sampler("ddim", 1000, steps, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall")
        assert torch.equal(s.timesteps, g[f"timesteps_{steps}"])
        assert torch.equal(s.sampling_sigmas, g[f"sampling_sigmas_{steps}"])
        assert torch.equal(s.min_var_coef, g[f"min_var_coef_{steps}"])
    assert torch.equal(s.sigmas, g["sigmas"]) and torch.equal(s.alphas_cumprod, g["alphas_cumprod"])
    assert s.timesteps[0].item() == 954 and s.timesteps[-1].item() == -1
    assert torch.equal(s.get_t_from_sigma(g["t_grid_sigma"]), g["t_grid_t"])         # ties -> left, > sigma_max -> 1000
    for sched in ("quadratic", "cosine", "sigmoid"):
        s2 = get_sampler("ddim", 1000, 20, beta_schedule=sched, sigma_style="DDIM", start_sigma=0, end_sigma=0)
        assert torch.equal(s2.sigmas, g[f"sigmas_{sched}"]) and torch.equal(s2.timesteps, g[f"timesteps_{sched}"])
    s3 = get_sampler("ddim", 1000, 10, sigma_style="DDIM", start_sigma=100, end_sigma=0, set_alpha_to_one=False)
    assert torch.equal(s3.timesteps, g["timesteps_noalpha1"]) and torch.equal(s3.sampling_sigmas, g["sampling_sigmas_noalpha1"])


def test_pred_xprev_all_samplers():
    from oracle.sched import SAMPLERS, get_sampler
    g = load_npz("sched")
    x0, xt, eps, learned, noise, st, sp = (g[k] for k in ("px_x0", "px_xt", "px_eps", "px_learned", "px_noise", "px_st", "px_sp"))
    for name in SAMPLERS:
        for var in ("fixedsmall", "fixedlarge", "learned"):
            for eta in (0.0, 0.85):
                s = get_sampler(name, 1000, 50, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var=var, eta=eta)
                lv = s.get_eps_logvar(st, sp, learned if var == "learned" else None)
                assert torch.equal(lv, g[f"lv_{var}"])
                xp = s.pred_xprev(x0=x0, eps=eps, sigma_t=st, sigma_prev=sp, xt=xt, log_variance=lv, noise=noise)
                assert torch.equal(xp, g[f"px_{name}_{var}_{eta}"]), (name, var, eta)


LOOPS = ["loop_simple_pred", "loop_simple_base", "loop_simple_partial", "loop_simple_orig_eta", "loop_simple_threshold",
         "loop_adm_dynamic", "loop_adm_eta", "loop_admb_ddpm"]


def _stats(a):
    return torch.stack([a.flatten(2).mean(-1), a.flatten(2).abs().mean(-1)], dim=-1)


def run_oracle_loop(g):
    from oracle.loop import DiffusionOracle
    from oracle.sched import get_sampler
    c = g["cfg"]
    eps_fn, enc_fn, sig_fn, _ = oracle_nets(c["tag"])
    s = get_sampler(c["sampler"], 1000, c["steps"], sigma_style="DDIM", start_sigma=c["start_sigma"], end_sigma=0,
                    sampler_var=c["var"], eta=c["eta"])
    o = DiffusionOracle(eps_fn, enc_fn, sig_fn, s, (3, c["res"], c["res"]), learn_epsvar=c["tag"] == "adm_tiny",
                        norm_min=c["norm_min"], norm_max=c["norm_max"], clip_fn=c["clip"])
    shape = (c["B"], 3, c["res"], c["res"])
    ng = torch.Generator().manual_seed(c["seed"] + 1)
    noises = [torch.randn(shape, generator=ng) for _ in range(int(g["n_noise"]))] or None
    trace = {}
    xT = g["z"] / (1 / (s.sampling_sigmas[0] ** 2 + 1)).sqrt()
    x = o.denoise_loop(shape, style=c["style"], norm_eps=c["norm_eps"], refine_prior_sigma=c["refine"], xT=xT,
                       sigma_pred_threshold=c["threshold"], trace=trace, noise_list=noises)
    return x, trace


@pytest.mark.parametrize("name", LOOPS)
def test_denoise_loop_matches_reference(name):
    g = load_npz(name)
    x, trace = run_oracle_loop(g)
    assert max_err(trace["x0"][0], g["x0_first"]) < 1e-5
    assert max_err(_stats(torch.stack(trace["x0"])), g["x0_stats"]) < 1e-4
    assert max_err(x, g["x"]) < 1e-4, max_err(x, g["x"])


@pytest.mark.parametrize("name", ["loop_edm_pred", "loop_edm_base", "loop_edm_euler", "loop_edm_cos", "loop_edm_p3"])
def test_edm_sampler_matches_reference(name):
    from oracle.loop import EdmOracle
    g = load_npz(name)
    c = g["cfg"]
    eps_fn, enc_fn, sig_fn, _ = oracle_nets("edm_tiny")
    o = EdmOracle(eps_fn, enc_fn, sig_fn, (3, 32, 32), num_timesteps=c["steps"], norm_min=0.0, norm_max=54.63,
                  S_churn=c["S_churn"])
    torch.manual_seed(3)
    x = o.edm_sampler(g["latents"], style=c["style"], norm_eps=c["norm_eps"], eps_ratio=0.5, eps_scale=c["eps_scale"],
                      use_second_order=c["second"])
    assert x.dtype == torch.float64
    assert max_err(x, g["x"]) < 1e-4, max_err(x, g["x"])


def test_continuous_t_trajectory_is_conditioned_at_the_1e_3_level():
    """How far does the REFERENCE arithmetic itself move when the initial state is perturbed by +-1 ulp per element?
    With continuous t the refined sigma feeds t through an interpolation whose slope dt/dsigma is 3 ... 1000, so rounding-level
    differences are amplified; measured here with the CPU oracle (bit-identical to the reference on this fixture): final-sample
    L-inf 3e-4 ... 1.1e-3 over four sign patterns.  This is the floor any independent implementation of the same arithmetic
    sits on - the GPU tests gate the continuous-t trajectories at 3e-3 (about 3x this floor) and everything that is not
    chaotically amplified (first step, sigma traces, all discrete-t fixtures) at the north-star 1e-3."""
    from oracle.loop import DiffusionOracle
    from oracle.sched import get_sampler
    g = load_npz("cont_linear")
    eps_fn, enc_fn, sig_fn, _ = oracle_nets("simple_tiny")
    kw = dict(sigma_style="Linear", start_sigma=100, end_sigma=0.01, sampler_var="fixedsmall", eta=0.0, continuous_t=True)

    def run(xT):
        s = get_sampler("ddim", 1000, 10, **kw)
        o = DiffusionOracle(eps_fn, enc_fn, sig_fn, s, (3, 32, 32), learn_epsvar=False, norm_min=0.0, norm_max=54.63, clip_fn="clamp")
        return o.denoise_loop((2, 3, 32, 32), style="pred", norm_eps=True, refine_prior_sigma=True, xT=xT, sigma_pred_threshold=960)

    s0 = get_sampler("ddim", 1000, 10, **kw)
    z = torch.randn((2, 3, 32, 32), generator=torch.manual_seed(1234))
    xT = (z / (1 / (s0.sampling_sigmas[0] ** 2 + 1)).sqrt()).float()
    x_ref = run(xT)
    assert max_err(x_ref, g["x"]) < 1e-4
    moved = []
    for seed in range(4):
        up = torch.rand(xT.shape, generator=torch.Generator().manual_seed(seed)) < 0.5
        xp = torch.where(up, torch.nextafter(xT, torch.full_like(xT, float("inf"))), torch.nextafter(xT, torch.full_like(xT, float("-inf"))))
        moved.append(max_err(run(xp), x_ref))
    print("continuous-t: final-sample L-inf of the oracle against itself under +-1 ulp input noise:", " ".join(f"{v:.1e}" for v in moved))
    assert max(moved) > 2e-4          # the amplification is real (if this ever drops, tighten the GPU gates accordingly)
    assert max(moved) < 3e-3          # ... and bounded: the 3e-3 GPU gate is not vacuous
