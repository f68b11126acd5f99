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

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_policy")]

LOOPS = ["loop_simple_pred", "loop_simple_base", "loop_simple_partial", "loop_simple_orig_eta", "loop_simple_threshold",
         "loop_adm_dynamic", "loop_adm_eta", "loop_admb_ddpm"]


def _stats(a):
    return torch.stack([a.flatten(2).mean(-1), a.flatten(2).abs().mean(-1)], dim=-1)


def run_hip_loop(g, dtype=torch.float32, return_log=True, graphs=False, matmul="native", chunk_size=1):
    from diffusion_nlc_amd.experiments import ImageExperiment
    from diffusion_nlc_amd.schedulers import get_sampler
    c = g["cfg"]
    eps, sig = _models(c["tag"], dtype, matmul)
    s = get_sampler(c["sampler"], 1000, c["steps"], sigma_style="DDIM", start_sigma=c["start_sigma"], end_sigma=0,
                    sampler_var=c["var"], eta=c["eta"])
    s.to("cuda:0")
    exp = ImageExperiment(eps, s, batch_size=c["B"], data_shape=(3, c["res"], c["res"]), seed=c["seed"], device="cuda:0")
    exp.set_model(eps, sig, learn_epsvar=c["tag"] == "adm_tiny")
    exp.set_norm_maxmin(c["norm_min"], c["norm_max"])
    exp.set_clip_fn(c["clip"])
    exp.use_graphs = graphs
    shape = (c["B"], 3, c["res"], c["res"])
    ng = torch.Generator().manual_seed(c["seed"] + 1)
    noises = [torch.randn(shape, generator=ng) for _ in range(int(g["n_noise"]))] or None
    gen = exp.new_gen()                       # same seed -> the same z the reference drew
    x, logs = exp.denoise_loop(shape=shape, gen=gen, style=c["style"], norm_eps=c["norm_eps"],
                               refine_prior_sigma=c["refine"], return_log=return_log, chunk_size=chunk_size,
                               sigma_pred_threshold=c["threshold"], noise_list=noises)
    return x, logs


def test_chunk_size_microbatches_the_eps_network():
    """``chunk_size`` (src/experiments.py:436-450): the eps network is evaluated on len(xt) // chunk_size samples at a time - a
    memory knob upstream.  No op of the path couples samples; what can differ is the kernel the dispatch picks for the smaller
    launch (tile counts decide), i.e. the order of f32 additions: the f32 sample agrees with the unchunked one to 1e-4 (and with the
    reference golden to the usual 1e-3), and a chunked run replayed from captured graphs equals the chunked eager run bit for bit."""
    g = load_npz("loop_adm_dynamic")
    a, _ = run_hip_loop(g, return_log=False, chunk_size=1)
    b, _ = run_hip_loop(g, return_log=False, chunk_size=2)
    c, _ = run_hip_loop(g, return_log=False, chunk_size=2, graphs=True)
    assert max_err(a, b) < 1e-4 and max_err(b, g["x"]) < 1e-3
    assert torch.equal(b, c)


@pytest.mark.parametrize("matmul", ["native", "f16x3"])
@pytest.mark.parametrize("name", LOOPS)
def test_f32_loop_matches_reference(name, matmul):
    g = load_npz(name)
    x, logs = run_hip_loop(g, matmul=matmul)
    assert max_err(logs[0][0], g["z"]) == 0.0                      # identical host-drawn start
    e0 = max_err(logs[3][0], g["x0_first"])
    es = max_err(_stats(torch.stack(logs[3])), g["x0_stats"])
    ex = max_err(x, g["x"])
    print(f"{name}: f32 ({matmul}) L-inf first x0 {e0:.2e}, stats {es:.2e}, final {ex:.2e}")
    assert e0 < 1e-3 and es < 1e-3 and ex < 1e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("name", ["loop_simple_pred", "loop_adm_dynamic"])
def test_16bit_loop_tracks_reference(name, dtype):
    """bf16 operands perturb every network output by ~1e-2 of its scale; through sigma=100 and the +-1 clamp the
    multi-step trajectory of a RANDOM-weight network is not comparable pixel-wise (reported, not gated).
    Gated: the first step (same input state) stays close on average and the run stays finite."""
    g = load_npz(name)
    x, logs = run_hip_loop(g, dtype=dtype, return_log=True)
    first = (logs[3][0].double() - g["x0_first"].double()).abs().mean().item()
    ex = max_err(x, g["x"])
    print(f"{name}: {dtype} first-step mean |dx0| {first:.2e}; final L-inf {ex:.2e} (informational)")
    # (builds whose f32-side kernels differ by 1-2 ulp move this statistic between 4.7e-2 and 5.4e-2 on the ADM
    #  fixture - dynamic thresholding divides by a per-sample quantile - so the gate is a loose 1e-1)
    assert torch.isfinite(x).all() and first < 1e-1


@pytest.mark.parametrize("name,dtype", [("loop_adm_dynamic", torch.float32), ("loop_adm_dynamic", torch.bfloat16),
                                        ("loop_simple_threshold", torch.float32), ("loop_admb_ddpm", torch.bfloat16),
                                        ("loop_adm_dynamic", torch.float16)])
def test_hipgraph_replay_is_bit_identical_to_eager_launches(name, dtype):
    """Network evaluations replayed from captured hipGraphs (one per entry point / shape; the threshold fixture switches
    between the NLC and the plain branch mid-run, the DDPM one injects host noise every step) against the same loop with
    every kernel launched eagerly: the same kernels on the same data -> equal bit for bit, and several replays deep."""
    g = load_npz(name)
    x_eager, _ = run_hip_loop(g, dtype=dtype, return_log=False)
    x_graph, _ = run_hip_loop(g, dtype=dtype, return_log=False, graphs=True)
    assert torch.equal(x_graph, x_eager)
    if dtype == torch.float32:
        assert max_err(x_graph, g["x"]) < 1e-3


def test_loop_without_logging_is_identical():
    g = load_npz("loop_simple_pred")
    x1, _ = run_hip_loop(g, return_log=True)
    x2, logs = run_hip_loop(g, return_log=False)
    assert torch.equal(x1, x2) and logs[1] == []


def test_nan_break_is_the_same_eager_and_deferred():
    """A NaN state ends the loop (src/experiments.py:389).  With logging the flag is read every step; without, one step
    late and without draining the queue - both must stop at the same step and return the same tensor."""
    from diffusion_nlc_amd.experiments import ImageExperiment
    from diffusion_nlc_amd.schedulers import get_sampler
    eps, sig = _models("simple_tiny", torch.float32)
    s = get_sampler("ddim", 1000, 6, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall", eta=0.0)
    s.to("cuda:0")
    exp = ImageExperiment(eps, s, batch_size=2, data_shape=(3, 32, 32), seed=3, device="cuda:0")
    exp.set_model(eps, sig, learn_epsvar=False)
    exp.set_norm_maxmin(0.0, 54.63)
    exp.set_clip_fn("none")
    xT = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(5)) * 100
    xT[1, 2, 7, 9] = float("nan")
    kw = dict(shape=(2, 3, 32, 32), xT=xT, style="base", norm_eps=False, refine_prior_sigma=False, chunk_size=1)
    xa, logs = exp.denoise_loop(return_log=True, **kw)
    xb, _ = exp.denoise_loop(return_log=False, **kw)
    assert len(logs[1]) == 1                                   # stopped after the first step
    assert torch.isnan(xa).any() and torch.equal(torch.isnan(xa), torch.isnan(xb))
    assert torch.equal(torch.nan_to_num(xa), torch.nan_to_num(xb))


@pytest.mark.parametrize("name", ["loop_edm_pred", "loop_edm_base", "loop_edm_euler", "loop_edm_cos", "loop_edm_p3"])
def test_f32_edm_sampler_matches_reference(name):
    """EDM / Heun + NLC (float64 state, float32 network) vs the reference's own edm_sampler output."""
    from diffusion_nlc_amd.experiments import EDMImageExperiment
    g = load_npz(name)
    c = g["cfg"]
    eps, sig = _models("edm_tiny", torch.float32)
    exp = EDMImageExperiment(eps, None, batch_size=2, data_shape=(3, 32, 32), seed=0, device="cuda:0", num_timesteps=c["steps"],
                             S_churn=c["S_churn"])
    exp.set_model(eps, sig, learn_epsvar=False)
    exp.set_norm_maxmin(0.0, 54.63)
    torch.manual_seed(3)                                  # the churn noise comes from the global host generator (:880)
    x = exp.edm_sampler(shape=(2, 3, 32, 32), latents=g["latents"], style=c["style"], norm_eps=c["norm_eps"],
                        eps_ratio=0.5, eps_scale=c["eps_scale"], use_second_order=c["second"])
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
    _, a = exp.evaluate_edm(4, None, style="pred_partial,pred", norm_eps="000", return_samples=True)
    _, b = exp.evaluate_edm(4, None, style="pred_partial,pred", norm_eps="000", return_samples=True)
    assert a.shape == (4, 3, 32, 32) and torch.equal(a, b) and a.min() >= 0 and a.max() <= 1
    with pytest.raises(ValueError):
        exp.evaluate_edm(3, None)


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


def _simple_experiment(sampler, B=2, res=32, seed=1234):
    from diffusion_nlc_amd.experiments import ImageExperiment
    eps, sig = _models("simple_tiny", torch.float32)
    sampler.to("cuda:0")
    exp = ImageExperiment(eps, sampler, batch_size=B, data_shape=(3, res, res), seed=seed, device="cuda:0")
    exp.set_model(eps, sig, learn_epsvar=False)
    exp.set_norm_maxmin(0.0, 54.63)
    exp.set_clip_fn("clamp")
    return exp


def test_continuous_t_loop_matches_reference():
    """'Linear' sigma spacing with continuous t (Interp1d lookups on the device)."""
    from diffusion_nlc_amd.schedulers import get_sampler
    g = load_npz("cont_linear")
    s = get_sampler("ddim", 1000, 10, sigma_style="Linear", start_sigma=100, end_sigma=0.01, sampler_var="fixedsmall", eta=0.0,
                    continuous_t=True)
    exp = _simple_experiment(s)
    x, logs = exp.denoise_loop(shape=(2, 3, 32, 32), gen=exp.new_gen(), style="pred", norm_eps=True, refine_prior_sigma=True,
                               return_log=True, chunk_size=1, sigma_pred_threshold=960)
    e0, ex = max_err(logs[3][0], g["x0_first"]), max_err(x, g["x"])
    print(f"continuous-t loop: L-inf first x0 {e0:.2e}, final {ex:.2e}")
    # With continuous t a 1-ulp difference in ||x|| moves t itself (slope d t / d sigma is 3 ... 1000) and this random-weight
    # net amplifies it over 10 steps: the REFERENCE arithmetic run against itself with the initial state perturbed by +-1 ulp
    # ends 3e-4 ... 1.1e-3 apart (tests/test_oracle_golden.py::test_continuous_t_trajectory_is_conditioned_at_the_1e_3_level).
    # Gate: first step at the north-star 1e-3; the 10-step trajectory at 3e-3 (about 3x that conditioning floor).
    assert e0 < 1e-3 and ex < 3e-3


@pytest.mark.parametrize("name", ["proj_linear", "proj_discrete"])
def test_projection_loop_matches_reference(name):
    """image_sample.projection_loop (per-sample sigma re-estimation on the device)."""
    from diffusion_nlc_amd.schedulers import get_sampler
    g = load_npz(name)
    c = g["cfg"]
    if name == "proj_linear":
        s = get_sampler("ddim", 1000, 10, sigma_style="Linear", start_sigma=100, end_sigma=0.01, sampler_var="fixedsmall", eta=0.0,
                        continuous_t=True)
        kw = dict(style="pred", refine_prior_sigma=True)
    else:
        s = get_sampler("ddim", 1000, 10, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall", eta=0.0)
        kw = dict(style="pred_partial", refine_prior_sigma=False)
    exp = _simple_experiment(s)
    x, logs = exp.projection_loop(shape=(2, 3, 32, 32), gen=exp.new_gen(), norm_eps=True, return_log=True, chunk_size=1,
                                  sigma_estimate_rate=c["rate"], sigma_pred_threshold=960, recal_sigma_prev=c["recal"], **kw)
    tr = torch.stack([s_.reshape(-1) for s_ in logs[4][1:]])
    es = ((tr.double() - g["sigma_trace"].double()).abs() / g["sigma_trace"].double().abs().clamp(min=1e-3)).max().item()
    e0, ex = max_err(logs[3][0], g["x0_first"]), max_err(x, g["x"])
    print(f"{name}: L-inf first x0 {e0:.2e}, final {ex:.2e}; sigma trace rel {es:.2e}")
    # trajectory tolerance: see test_continuous_t_loop_matches_reference (the discrete-t fixture stays at ~2e-4)
    assert e0 < 1e-3 and es < 1e-3 and ex < (3e-3 if name == "proj_linear" else 1e-3)
    x2, _ = exp.projection_loop(shape=(2, 3, 32, 32), gen=exp.new_gen(), norm_eps=True, return_log=False, chunk_size=1,
                                sigma_estimate_rate=c["rate"], sigma_pred_threshold=960, recal_sigma_prev=c["recal"], **kw)
    assert torch.equal(x, x2)


def test_projection_with_redesigned_sigmas_and_inpainting():
    """The paper's 'pred_proj' mode: redesigned sigma tail (image_sample.py:788-800) + projection_loop + inpainting
    constraint, including the reference driver's early stop on an exactly satisfied constraint."""
    from functools import partial
    from diffusion_nlc_amd.constraint_functions import Constraint_Function, Inpainting
    from diffusion_nlc_amd.schedulers import get_sampler, redesign_sigma
    g = load_npz("proj_redesign")
    c = g["cfg"]
    s = get_sampler("ddim", 1000, c["num_timesteps"], sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall", eta=0.0)
    redesign_sigma(s, c["num_timesteps"], c["max_T"], c["cycle_size"], c["min_sigma"], c["max_sigma"], c["sigma_gamma"])
    exp = _simple_experiment(s)
    op = Inpainting(3, 32, g["missing"], "cuda:0")
    y = op.A(g["x_gt"])
    cf = Constraint_Function("inpainting_random", op, channels=3, image_size=32)
    common = dict(shape=(2, 3, 32, 32), style="pred", norm_eps=True, refine_prior_sigma=True, chunk_size=1,
                  sigma_estimate_rate=c["rate"], max_T=c["max_T"], sigma_pred_threshold=960, recal_sigma_prev=True,
                  constrain_loss=partial(cf.loss, y=y))
    x, logs = exp.projection_loop(gen=exp.new_gen(), constrain_fn=partial(cf.constraint_fn, y=y), return_log=True,
                                  stop_condition=-1.0, **common)
    assert len(logs[1]) == int(g["n_steps"])
    tr = torch.stack([s_.reshape(-1) for s_ in logs[4][1:]])
    es = ((tr.double() - g["sigma_trace"].double()).abs() / g["sigma_trace"].double().abs().clamp(min=1e-3)).max().item()
    e0, el = max_err(logs[3][0], g["x0_first"]), max_err(logs[3][-1], g["x0_last"])
    ec = (torch.stack(logs[5]).double() - g["const_loss"].double()).abs().max().item()
    per_step = (torch.stack(logs[3])[..., ::4, ::4].double() - g["x0_sub"].double()).abs().flatten(1).max(dim=1).values
    print(f"proj_redesign: L-inf first x0 {e0:.2e}, last x0 {el:.2e}; sigma trace rel {es:.2e}; const-loss abs {ec:.2e}")
    print("  per-step L-inf of x0 (subsampled):", " ".join(f"{v:.1e}" for v in per_step.tolist()))
    # the RETURNED sample (north-star quantity) is checked at 1e-3 below.  The logged mid-trajectory x0 of the
    # unknown pixels of this random-weight net amplifies f32 summation-order noise ~10x between steps 1 and 4
    # and then stays flat (measured 1e-4 -> 1e-3 -> 7e-4): gate the trajectory at 3e-3, everything else at 1e-3.
    assert e0 < 1e-3 and es < 1e-3 and ec < 1e-3 and el < 3e-3 and per_step.max().item() < 3e-3
    first_min = lambda v: int((v == v.min()).nonzero()[0])
    if first_min(torch.stack(logs[5]).mean(1)) == first_min(g["const_loss"].mean(1)):   # same winner -> same sample
        assert max_err(x, g["x"]) < 1e-3
    # best-x0 selection compares constraint losses that are rounding residue (0 .. 2e-6): whichever step wins,
    # the returned sample must be one of the logged post-constraint x0 and satisfy the constraint
    assert any(torch.equal(x, p) for p in logs[3])
    # fused projection inside nlc_sched_step + the driver's stop_condition=0.0
    x2, logs2 = exp.projection_loop(gen=exp.new_gen(), constrain_fn=cf.bind(y, (2, 3, 32, 32)), return_log=False,
                                    stop_condition=0.0, **common)
    assert max_err(x2, g["x"]) < 1e-3
