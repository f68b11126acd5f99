"""BASELINE config 2 at its FULL model size (ADM-256, 552.8 M + 61.4 M parameters, 256x256, bf16 - the benchmarked path),
through size-independent properties: the CPU oracle needs ~25 s per NLC step per image there, so the checks are
determinism, per-sample independence (batch-permutation equivariance - no op on the path couples samples,
SURVEY.md §8e), finiteness / range, and agreement of the bf16 first step with the f32 path of the same kernels."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def adm256():
    import bench
    dev = torch.device("cuda:0")
    return bench.make_experiment(dict(bench.ADM256), dev, torch.bfloat16, 4, 2)


def _run(exp, xT):
    x, _ = exp.denoise_loop(shape=tuple(xT.shape), xT=xT, style="pred", norm_eps=True, refine_prior_sigma=True,
                            return_log=False, chunk_size=1, sigma_pred_threshold=960)
    return x


def test_adm256_bf16_is_deterministic_and_sample_independent(adm256):
    exp = adm256
    g = torch.Generator().manual_seed(1234)
    sigma0 = exp.scheduler.sampling_sigmas[0]
    xT = (torch.randn(4, 3, 256, 256, generator=g) / (1 / (sigma0 ** 2 + 1)).sqrt()).to("cuda:0")
    a = _run(exp, xT)
    b = _run(exp, xT)
    assert torch.equal(a, b)                                         # fixed-order reductions everywhere: bit-reproducible
    perm = torch.tensor([2, 0, 3, 1])
    c = _run(exp, xT[perm].contiguous())
    assert torch.equal(c, a[perm])                                   # samples never see each other
    assert torch.isfinite(a).all() and a.abs().max() <= 1.0 + 1e-6   # dynamic-threshold clip keeps x0 in [-1, 1]


def test_adm256_bf16_first_step_tracks_f32(adm256):
    """One network evaluation at full size: bf16 (split-K, ride-along GroupNorm statistics, matrix-core first layer) against
    the f32 path of the same kernels (no split-K, Chan-merged statistics, exact-f32 MFMA): per-op bf16 error is ~1e-2 of
    scale, so the eps prediction must agree to a few percent of its scale."""
    exp = adm256
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(2, 3, 256, 256, generator=g) * 30).to("cuda:0")
    t = torch.tensor([700.0, 321.0], device="cuda:0")
    c_in = torch.tensor([0.03, 0.05], device="cuda:0")
    out_bf = exp.model.run(x, t, mode="forward", in_scale=c_in).clone()
    exp.model.set_compute_dtype(torch.float32)
    try:
        out_f32 = exp.model.run(x, t, mode="forward", in_scale=c_in)
    finally:
        exp.model.set_compute_dtype(torch.bfloat16)
    scale = out_f32.abs().max().item()
    err = (out_bf - out_f32).abs().max().item()
    rel_rms = ((out_bf - out_f32).pow(2).mean().sqrt() / out_f32.pow(2).mean().sqrt()).item()
    print(f"ADM-256 forward: bf16 vs f32 L-inf {err:.3e} (scale {scale:.3e}), relative RMS {rel_rms:.3e}")
    assert torch.isfinite(out_bf).all() and err <= 6e-2 * scale and rel_rms <= 2e-2


def test_adm256_f32_two_steps_match_the_oracle_at_full_size(adm256):
    """The headline model itself (ADM-256, 614 M parameters, 256x256) in f32 against the CPU oracle: two full
    DDIM+NLC timesteps (refine -> encode -> sigma net -> corrected sigma / t -> eps forward -> learned variance,
    dynamic-threshold clip -> scheduler update) for one image; per-pixel L-inf <= 1e-3 (the north-star tolerance).
    ~1.5 s of oracle time per timestep on the GPU box's 16 host cores."""
    import bench
    from diffusion_nlc_amd.filler import fill_state_dict
    from diffusion_nlc_amd.script_util import create_sigma_eps_model
    from oracle import adm
    from oracle.loop import DiffusionOracle
    from oracle.sched import get_sampler as oracle_sampler
    exp = adm256
    cfg = dict(bench.ADM256)
    ucfg, scfg, _ = adm.configs_from_factory(**cfg)
    eps_m, sig_m, _ = create_sigma_eps_model(**cfg)
    sd_e = fill_state_dict(eps_m.state_dict(), seed=0)                    # the same filler weights the GPU models carry
    sd_s = fill_state_dict(sig_m.state_dict(), seed=1, overrides=bench.SIGMA_OVERRIDES)
    del eps_m, sig_m
    kw = dict(sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="learned", eta=0.0)
    osched = oracle_sampler("ddim", 1000, 2, **kw)
    o = DiffusionOracle(lambda x, t: adm.unet(sd_e, ucfg, x, t, "forward"), lambda x, t: adm.unet(sd_e, ucfg, x, t, "encode"),
                        lambda f: adm.sigma_net(sd_s, scfg, f), osched, (3, 256, 256), learn_epsvar=True, norm_min=0.0,
                        norm_max=440.0, clip_fn="dynamic")
    z = torch.randn((1, 3, 256, 256), generator=torch.Generator().manual_seed(99))
    xT = z / (1 / (osched.sampling_sigmas[0] ** 2 + 1)).sqrt()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    x_cpu = o.denoise_loop((1, 3, 256, 256), style="pred", norm_eps=True, refine_prior_sigma=True, xT=xT, sigma_pred_threshold=960)
    exp.model.set_compute_dtype(torch.float32)
    exp.sigma_model.set_compute_dtype(torch.float32)
    try:
        x_gpu, _ = exp.denoise_loop(shape=(1, 3, 256, 256), xT=xT, style="pred", norm_eps=True, refine_prior_sigma=True,
                                    return_log=False, chunk_size=1, sigma_pred_threshold=960)
    finally:
        exp.model.set_compute_dtype(torch.bfloat16)
        exp.sigma_model.set_compute_dtype(torch.bfloat16)
    err = (x_gpu.double() - x_cpu.double()).abs().max().item()
    print(f"ADM-256 f32, 2 DDIM+NLC timesteps, 1 image: HIP vs CPU oracle L-inf = {err:.3e}")
    assert err <= 1e-3, err
