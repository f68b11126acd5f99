"""DDIM + NLC sampling loop on the GPU (HIP networks + sampler kernels through the C ABI) against
the golden outputs of the reference's own ``denoise_loop`` and against the CPU oracle.

f32 path: final sample L-inf <= 1e-3 (the north-star tolerance) on every loop fixture.
bf16 path: reported, gated loosely (the NLC feedback through searchsorted is discontinuous).
"""
import json

import pytest
import torch

from tests.test_host_cpu import build_product
from tests.test_nets_gpu import _models
from tests.util import load_npz, max_err

pytestmark = pytest.mark.gpu

LOOPS = ["loop_simple_pred", "loop_simple_base", "loop_simple_partial", "loop_simple_orig_eta", "loop_simple_threshold",
         "loop_adm_dynamic", "loop_adm_eta", "loop_admb_ddpm"]


def _stats(a):
    return torch.stack([a.flatten(2).mean(-1), a.flatten(2).abs().mean(-1)], dim=-1)


def run_hip_loop(g, dtype=torch.float32, return_log=True):
    from diffusion_nlc_amd.experiments import ImageExperiment
    from diffusion_nlc_amd.schedulers import get_sampler
    c = g["cfg"]
    eps, sig = _models(c["tag"], dtype)
    s = get_sampler(c["sampler"], 1000, c["steps"], sigma_style="DDIM", start_sigma=c["start_sigma"], end_sigma=0,
                    sampler_var=c["var"], eta=c["eta"])
    s.to("cuda:0")
    exp = ImageExperiment(eps, s, batch_size=c["B"], data_shape=(3, c["res"], c["res"]), seed=c["seed"], device="cuda:0")
    exp.set_model(eps, sig, learn_epsvar=c["tag"] == "adm_tiny")
    exp.set_norm_maxmin(c["norm_min"], c["norm_max"])
    exp.set_clip_fn(c["clip"])
    shape = (c["B"], 3, c["res"], c["res"])
    ng = torch.Generator().manual_seed(c["seed"] + 1)
    noises = [torch.randn(shape, generator=ng) for _ in range(int(g["n_noise"]))] or None
    gen = exp.new_gen()                       # same seed -> the same z the reference drew
    x, logs = exp.denoise_loop(shape=shape, gen=gen, style=c["style"], norm_eps=c["norm_eps"],
                               refine_prior_sigma=c["refine"], return_log=return_log, chunk_size=1,
                               sigma_pred_threshold=c["threshold"], noise_list=noises)
    return x, logs


@pytest.mark.parametrize("name", LOOPS)
def test_f32_loop_matches_reference(name):
    g = load_npz(name)
    x, logs = run_hip_loop(g)
    assert max_err(logs[0][0], g["z"]) == 0.0                      # identical host-drawn start
    e0 = max_err(logs[3][0], g["x0_first"])
    es = max_err(_stats(torch.stack(logs[3])), g["x0_stats"])
    ex = max_err(x, g["x"])
    print(f"{name}: f32 L-inf first x0 {e0:.2e}, stats {es:.2e}, final {ex:.2e}")
    assert e0 < 1e-3 and es < 1e-3 and ex < 1e-3


@pytest.mark.parametrize("name", ["loop_simple_pred", "loop_adm_dynamic"])
def test_bf16_loop_tracks_reference(name):
    """bf16 operands perturb every network output by ~1e-2 of its scale; through sigma=100 and the +-1 clamp the
    multi-step trajectory of a RANDOM-weight network is not comparable pixel-wise (reported, not gated).
    Gated: the first step (same input state) stays close on average and the run stays finite."""
    g = load_npz(name)
    x, logs = run_hip_loop(g, dtype=torch.bfloat16, return_log=True)
    first = (logs[3][0].double() - g["x0_first"].double()).abs().mean().item()
    ex = max_err(x, g["x"])
    print(f"{name}: bf16 first-step mean |dx0| {first:.2e}; final L-inf {ex:.2e} (informational)")
    assert torch.isfinite(x).all() and first < 5e-2


def test_loop_without_logging_is_identical():
    g = load_npz("loop_simple_pred")
    x1, _ = run_hip_loop(g, return_log=True)
    x2, logs = run_hip_loop(g, return_log=False)
    assert torch.equal(x1, x2) and logs[1] == []


@pytest.mark.parametrize("name", ["loop_edm_pred", "loop_edm_base", "loop_edm_euler"])
def test_f32_edm_sampler_matches_reference(name):
    """EDM / Heun + NLC (float64 state, float32 network) vs the reference's own edm_sampler output."""
    from diffusion_nlc_amd.experiments import EDMImageExperiment
    g = load_npz(name)
    c = g["cfg"]
    eps, sig = _models("edm_tiny", torch.float32)
    exp = EDMImageExperiment(eps, None, batch_size=2, data_shape=(3, 32, 32), seed=0, device="cuda:0", num_timesteps=c["steps"])
    exp.set_model(eps, sig, learn_epsvar=False)
    exp.set_norm_maxmin(0.0, 54.63)
    x = exp.edm_sampler(shape=(2, 3, 32, 32), latents=g["latents"], style=c["style"], norm_eps=c["norm_eps"],
                        eps_ratio=0.5, eps_scale=1.0, use_second_order=c["second"])
    assert x.dtype == torch.float64
    ex = max_err(x.cpu(), g["x"])
    print(f"{name}: f32-net / f64-state L-inf {ex:.2e}")
    assert ex < 1e-3


def test_edm_evaluate_runs_and_is_deterministic():
    from diffusion_nlc_amd.experiments import EDMImageExperiment
    eps, sig = _models("edm_tiny", torch.float32)
    exp = EDMImageExperiment(eps, None, batch_size=2, data_shape=(3, 32, 32), seed=0, device="cuda:0", num_timesteps=4)
    exp.set_model(eps, sig, learn_epsvar=False)
    exp.set_norm_maxmin(0.0, 54.63)
    _, a = exp.evaluate_edm(4, style="pred_partial,pred", norm_eps="000")
    _, b = exp.evaluate_edm(4, style="pred_partial,pred", norm_eps="000")
    assert a.shape == (4, 3, 32, 32) and torch.equal(a, b) and a.min() >= 0 and a.max() <= 1
    with pytest.raises(ValueError):
        exp.evaluate_edm(3)


def test_inpainting_operator_and_constrained_loop():
    """SURVEY §8 f-1 / BASELINE config 4 (reduced): Inpainting.A / A_pinv and a constrained DDIM+NLC loop vs the reference."""
    from functools import partial
    from diffusion_nlc_amd.constraint_functions import Constraint_Function, Inpainting
    from diffusion_nlc_amd.experiments import ImageExperiment
    from diffusion_nlc_amd.schedulers import get_sampler
    g = load_npz("inpaint")
    c = g["cfg"]
    res, B = c["res"], c["B"]
    op = Inpainting(3, res, g["missing"], "cuda:0")
    y = op.A(g["x_gt"])
    assert torch.equal(y.cpu(), g["y"])                                            # pure gather: bit exact
    assert torch.equal(op.A_pinv(y).view(B, 3, res, res).cpu(), g["apy"])
    cf = Constraint_Function("inpainting_random", op, channels=3, image_size=res)
    eps, sig = _models("simple_tiny", torch.float32)
    s = get_sampler("ddim", 1000, c["steps"], sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall", eta=0.0)
    s.to("cuda:0")
    exp = ImageExperiment(eps, s, batch_size=B, data_shape=(3, res, res), seed=c["seed"], device="cuda:0")
    exp.set_model(eps, sig, learn_epsvar=False)
    exp.set_norm_maxmin(0.0, 54.63)
    exp.set_clip_fn("clamp")
    shape = (B, 3, res, res)
    # (1) reference-shaped call: generic callable + per-step constraint loss (logging on)
    x1, logs = exp.denoise_loop(shape=shape, gen=exp.new_gen(), style="pred", constrain_fn=partial(cf.constraint_fn, y=y),
                                norm_eps=True, refine_prior_sigma=True, return_log=True, chunk_size=1,
                                constrain_loss=partial(cf.loss, y=y), sigma_pred_threshold=960)
    e1 = max_err(x1, g["x"])
    # after the projection the forward loss is pure rounding residue (~1e-5): compare absolutely
    el = (torch.stack(logs[4]).double() - g["const_loss"].double()).abs().max().item()
    # (2) fused path: the projection runs inside nlc_sched_step
    x2, _ = exp.denoise_loop(shape=shape, gen=exp.new_gen(), style="pred", constrain_fn=cf.bind(y, shape), norm_eps=True,
                             refine_prior_sigma=True, return_log=False, chunk_size=1, constrain_loss=partial(cf.loss, y=y),
                             sigma_pred_threshold=960)
    e2 = max_err(x2, g["x"])
    print(f"inpaint: L-inf generic {e1:.2e}, fused {e2:.2e}, const-loss abs {el:.2e}")
    assert e1 < 1e-3 and e2 < 1e-3 and el < 1e-3
    known = g["apy"][:, :, :, :]
    mask = op.mask_chw.view(3, res, res).cpu().bool()
    assert torch.equal(x2[:, mask], known[:, mask])                                # known pixels are copied exactly
