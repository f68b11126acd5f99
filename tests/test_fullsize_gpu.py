"""BASELINE configs 2, 3 and 4 at their FULL model sizes against the CPU oracle, in every precision bench.py can run
(bench.PRECISIONS): f32 (exact f32 MFMA) and f32x3 (f32 storage, split-f16 three-pass matrix math) carry the north-star's
1e-3 per-pixel gate end to end; bf16 (the benchmarked dtype) and f16 (the reference's own use_fp16 mode) are gated on the first
timestep's x0 and on the NLC-corrected sigma, at ~1.5x what was measured on MI355X (regression tripwires, the measured values
are in the comments and in profiles/r03_summary.md).  Plus size-independent properties of the benchmarked path: determinism,
per-sample independence (batch-permutation equivariance - no op on the path couples samples), range."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_policy")]


def _set_precision(exp, name):
    import bench
    for m in (exp.model, exp.sigma_model):
        bench.set_precision(m, bench.PRECISIONS[name])


@pytest.fixture(scope="module")
def adm256():
    import bench
    dev = torch.device("cuda:0")
    return bench.make_experiment(dict(bench.ADM256), dev, bench.PRECISIONS["bf16"], 4, 2)


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


@pytest.fixture(scope="module")
def adm256_oracle():
    """The CPU oracle on the headline model itself (ADM-256, 614 M parameters, 256x256): two full DDIM+NLC timesteps of one
    image (~1.5 s of oracle time per timestep on the GPU box's 16 host cores).  Returns (xT, final sample, per-step trace)."""
    import bench
    from diffusion_nlc_amd.filler import fill_state_dict
    from diffusion_nlc_amd.script_util import create_sigma_eps_model
    from oracle import adm
    from oracle.loop import DiffusionOracle
    from oracle.sched import get_sampler as oracle_sampler
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
    trace = {}
    x_cpu = o.denoise_loop((1, 3, 256, 256), style="pred", norm_eps=True, refine_prior_sigma=True, xT=xT, sigma_pred_threshold=960,
                           trace=trace)
    return xT, x_cpu, trace


@pytest.mark.parametrize("prec", ["f32", "f32x3"])
def test_adm256_f32_two_steps_match_the_oracle_at_full_size(adm256, adm256_oracle, prec):
    """The headline model against the CPU oracle: two full DDIM+NLC timesteps (refine -> encode -> sigma net ->
    corrected sigma / t -> eps forward -> learned variance, dynamic-threshold clip -> scheduler update) for one image;
    per-pixel L-inf <= 1e-3 (the north-star tolerance), in exact f32 and in the split-f16 matrix mode (f32 storage, three
    16-bit MFMA passes per product - the path that carries the gate at a usable speed).  Runs under the production dispatch and
    with the halo kernel forced."""
    exp = adm256
    xT, x_cpu, _ = adm256_oracle
    _set_precision(exp, prec)
    try:
        x_gpu, _ = exp.denoise_loop(shape=(1, 3, 256, 256), xT=xT, style="pred", norm_eps=True, refine_prior_sigma=True,
                                    return_log=False, chunk_size=1, sigma_pred_threshold=960)
    finally:
        _set_precision(exp, "bf16")
    err = (x_gpu.double() - x_cpu.double()).abs().max().item()
    print(f"ADM-256 {prec}, 2 DDIM+NLC timesteps, 1 image: HIP vs CPU oracle L-inf = {err:.3e}")
    assert err <= 1e-3, err


# bf16 (the benchmarked dtype) against the ORACLE at full size.  Gates are what the measurement supports:
# the first timestep's x0 - one encode + sigma net + forward through bf16 convolutions, everything else f32 - and the
# sigma the NLC net corrected; a multi-step bf16 trajectory of a random-weight network is not pixel-comparable
# (discontinuous sigma -> t lookup), which is why the 1e-3 gate is carried by the f32 path above.
# Measured on MI355X (round 2): x0 L-inf 0.39, RMS 0.061 against an x0 RMS of 0.80 - at sigma_0 = 100 the first x0 is
# xt - 100 * eps, the difference of two ~100-sized tensors, so the ~0.9 % bf16 error of eps (test above) is amplified ~10x
# before the dynamic-threshold normalisation.  Gated at ~1.5x the measured values.
# (x0 L-inf, x0 RMS, sigma relative) gates per 16-bit type, ~1.5x the round-3 measurements on MI355X:
#   bf16  L-inf 0.041 / RMS 6.3e-3 / sigma 2.3e-3 under the production dispatch, 0.40 / 0.061 / 5.3e-3 with the halo kernel forced
#   f16   L-inf 5.6e-3 / RMS 7.3e-4 / sigma 2.7e-4 (both dispatches): 11 significand bits against bf16's 8
GATES_16 = {"bf16": (0.6, 0.09, 0.01), "f16": (1e-2, 1.2e-3, 1e-3)}


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_adm256_16bit_first_step_against_the_oracle_at_full_size(adm256, adm256_oracle, prec):
    exp = adm256
    xT, _, trace = adm256_oracle
    x0_cpu = trace["x0"][0]                                              # post-clip x0 of timestep 0, in [-1, 1]
    _set_precision(exp, prec)
    try:
        _, logs = exp.denoise_loop(shape=(1, 3, 256, 256), xT=xT, style="pred", norm_eps=True, refine_prior_sigma=True,
                                   return_log=True, chunk_size=1, sigma_pred_threshold=960)
        sig0 = exp.sigma_trace[0].double()                               # the corrected sigma of timestep 0 (what NLC is for)
    finally:
        _set_precision(exp, "bf16")
    x0_gpu = logs[3][0]
    d = (x0_gpu.double() - x0_cpu.double())
    linf, rms = d.abs().max().item(), d.pow(2).mean().sqrt().item()
    ref_rms = x0_cpu.double().pow(2).mean().sqrt().item()
    rel = ((sig0 - trace["sigma_t"][0].double()).abs() / trace["sigma_t"][0].double()).max().item()
    print(f"ADM-256 {prec}, first DDIM+NLC timestep, 1 image, x0 vs CPU oracle: L-inf {linf:.3e}, RMS {rms:.3e} (x0 RMS {ref_rms:.3e}); "
          f"NLC-corrected sigma: relative error {rel:.3e}")
    g = GATES_16[prec]
    assert torch.isfinite(x0_gpu).all() and linf <= g[0] and rms <= g[1] and rel <= g[2]


# cfg 4 / cfg 3 gates: f32 and f32x3 the north-star's 1e-3 per-pixel L-inf on the final sample.  16-bit types: at sigma_0 = 100 the
# first x0 = xt - 100 eps turns a 1 % error of eps into an O(1) error of x0, and the +-1 clamp then saturates single pixels to the
# opposite bound, so per-pixel L-inf says nothing there; gated are the RMS of the final sample and of the first timestep's x0, and
# the NLC-corrected sigma (cfg 4), resp. the RMS of the final sample (cfg 3) - tripwires at ~1.5x the values measured on MI355X.
# Measured (round 3): cfg 4 f32 5.4e-4, f32x3 2.7e-4 (L-inf); bf16 final RMS 0.074, first-x0 RMS 0.030, sigma 9.5e-4; f16 final RMS
# 3.2e-3 ... 6.9e-2 (the two dispatches part ways at a clamp within three timesteps), first-x0 RMS 4.1e-3, sigma 1.8e-4.
# cfg 3 f32 9.8e-6, f32x3 1.1e-5 (L-inf); bf16 RMS 5.4e-3, f16 6.8e-4 (sample RMS 1.22).
# The corrected sigma of ONE image is one draw of a scalar whose 16-bit error spreads 5x from image to image and from one summation
# order to the next (profiles/r03_sigma_precision.txt: bf16 0.6 ... 3.0e-3, f16 0.4 ... 3.2e-4 over six images; this test's own value
# moved 9.5e-4 -> 2.2e-3 when the split-K hand-off and the statistics reduction changed their summation order): its gate is the top of
# that spread x 1.5, not 1.5 x one draw.
CELEBA_GATES = {"f32": 1e-3, "f32x3": 1e-3, "bf16": (0.11, 0.045, 4.5e-3), "f16": (0.11, 6.5e-3, 5e-4)}
EDM_GATES = {"f32": 1e-3, "f32x3": 1e-3, "bf16": 8e-3, "f16": 1.1e-3}


@pytest.mark.parametrize("prec", ["f32", "f32x3", "bf16", "f16"])
def test_celebahq256_inpainting_matches_the_oracle_at_full_size(prec):
    """BASELINE config 4 at full model size: the DDPM 'simple' UNet (ch 128, mult 1-1-2-2-4-4, 113.7 M + 15.5 M
    parameters) at 256x256, seeded random 50 % inpainting mask, three DDIM+NLC timesteps with the projection fused into the
    scheduler kernel - HIP path vs the CPU oracle with the reference-shaped affine projection."""
    import argparse
    import bench
    from diffusion_nlc_amd import script_util
    from diffusion_nlc_amd.constraint_functions import Constraint_Function, Inpainting
    from diffusion_nlc_amd.experiments import ImageExperiment
    from diffusion_nlc_amd.filler import fill_state_dict
    from diffusion_nlc_amd.schedulers import get_sampler
    from oracle import simple
    from oracle.loop import DiffusionOracle
    from oracle.sched import get_sampler as oracle_sampler
    ns = argparse.Namespace
    mc = dict(ch=128, out_ch=3, ch_mult=[1, 1, 2, 2, 4, 4], num_res_blocks=2, attn_resolutions=[16], dropout=0.0, in_channels=3,
              resamp_with_conv=True, feat_layer=1, type="simple", sigma_block=2, sigma_dropout=0.0)
    config = ns(model=ns(**mc), data=ns(image_size=256), diffusion=ns(num_diffusion_timesteps=1000))
    eps, sig, _ = script_util.create_simple_sigma_eps_model(config)
    sd_e = fill_state_dict(eps.state_dict(), seed=0)
    sd_s = fill_state_dict(sig.state_dict(), seed=1, overrides={"final_mlp.weight": 0.1, "final_mlp.bias": 0.5})
    eps.load_state_dict(sd_e); sig.load_state_dict(sd_s)
    eps.to("cuda:0"); sig.to("cuda:0")
    res, B, steps = 256, 1, 3
    g = torch.Generator().manual_seed(11)
    missing_r = torch.randperm(res * res, generator=g)[: res * res // 2].long() * 3
    missing = torch.cat([missing_r, missing_r + 1, missing_r + 2], dim=0)
    x_gt = torch.rand(B, 3, res, res, generator=g) * 2 - 1
    kw = dict(sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall", eta=0.0)
    # ---- oracle (CPU): mask semantics of functions/svd_operators.py:324-359 restated on flat (c,h,w)-interleaved indices
    keep = torch.ones(3 * res * res, dtype=torch.bool); keep[missing] = False
    def to_flat(x): return x.reshape(B, 3, -1).permute(0, 2, 1).reshape(B, -1)          # pixel-major, channel-minor
    def from_flat(v): return v.reshape(B, -1, 3).permute(0, 2, 1).reshape(B, 3, res, res)
    y_flat = to_flat(x_gt)[:, keep]
    def constrain(x0):
        f = to_flat(x0).clone(); f[:, keep] = y_flat
        return from_flat(f)
    cfg = simple.SimpleConfig(ch=128, out_ch=3, ch_mult=(1, 1, 2, 2, 4, 4), num_res_blocks=2, attn_resolutions=(16,), in_channels=3,
                              resolution=256, resamp_with_conv=True, feat_layer=1, sigma_block=2)
    _, dim = simple.sigma_dims(cfg)
    osched = oracle_sampler("ddim", 1000, steps, **kw)
    o = DiffusionOracle(lambda x, t: simple.unet(sd_e, cfg, x, t, "forward"), lambda x, t: simple.unet(sd_e, cfg, x, t, "encode"),
                        lambda f: simple.sigma_net(sd_s, dim, cfg.sigma_block, f), osched, (3, res, res), learn_epsvar=False,
                        norm_min=0.0, norm_max=397.0, clip_fn="clamp")
    z = torch.randn((B, 3, res, res), generator=torch.Generator().manual_seed(5))
    xT = z / (1 / (osched.sampling_sigmas[0] ** 2 + 1)).sqrt()
    trace = {}
    x_cpu = o.denoise_loop((B, 3, res, res), style="pred", constrain_fn=constrain, norm_eps=True, refine_prior_sigma=True, xT=xT,
                           sigma_pred_threshold=960, trace=trace)
    # ---- HIP path
    bench.set_precision(eps, bench.PRECISIONS[prec])
    bench.set_precision(sig, bench.PRECISIONS[prec])
    s = get_sampler("ddim", 1000, steps, **kw)
    s.to("cuda:0")
    exp = ImageExperiment(eps, s, batch_size=B, data_shape=(3, res, res), seed=5, device="cuda:0")
    exp.set_model(eps, sig, learn_epsvar=False)
    exp.set_norm_maxmin(0.0, 397.0)
    exp.set_clip_fn("clamp")
    op = Inpainting(3, res, missing, "cuda:0")
    cf = Constraint_Function("inpainting_random", op, channels=3, image_size=res)
    y = op.A(x_gt)
    x_gpu, _ = exp.denoise_loop(shape=(B, 3, res, res), xT=xT, style="pred", constrain_fn=cf.bind(y, (B, 3, res, res)), norm_eps=True,
                                refine_prior_sigma=True, return_log=False, chunk_size=1, sigma_pred_threshold=960)
    # (logging switches the projection from the fused scheduler kernel to the reference-shaped affine form: a second, one-timestep run)
    _, logs = exp.denoise_loop(shape=(B, 3, res, res), xT=xT, style="pred", constrain_fn=cf.bind(y, (B, 3, res, res)), norm_eps=True,
                               refine_prior_sigma=True, return_log=True, chunk_size=1, sigma_pred_threshold=960, max_steps=1)
    err = (x_gpu.double() - x_cpu.double()).abs().max().item()
    known = (x_gpu - x_gt).abs()[:, keep.view(res * res, 3).t().reshape(3, res, res)].max().item()
    d0 = logs[3][0].double() - trace["x0"][0].double()
    first, first_rms = d0.abs().max().item(), d0.pow(2).mean().sqrt().item()
    rms = (x_gpu.double() - x_cpu.double()).pow(2).mean().sqrt().item()
    srel = ((exp.sigma_trace[0].double() - trace["sigma_t"][0].double()).abs() / trace["sigma_t"][0].double()).max().item()
    print(f"CelebA-HQ-256 simple UNet {prec}, 3 constrained DDIM+NLC timesteps: HIP vs CPU oracle L-inf = {err:.3e}, RMS {rms:.3e} (first "
          f"timestep's x0: L-inf {first:.3e}, RMS {first_rms:.3e}; its corrected sigma {srel:.3e} relative); known pixels off by {known:.1e}")
    assert known == 0.0
    g = CELEBA_GATES[prec]
    if isinstance(g, tuple):
        assert rms <= g[0] and first_rms <= g[1] and srel <= g[2]
    else:
        assert err <= g


@pytest.mark.parametrize("prec", ["f32", "f32x3", "bf16", "f16"])
def test_edm_cifar10_matches_the_oracle_at_full_size(prec):
    """BASELINE config 3 at full model size: SongUNet (128 channels, mult 2-2-2, 4 blocks, 55.7 M + 3.9 M parameters),
    32x32, Heun + NLC 'pred_partial,pred', 6 sigma steps, float64 state: HIP vs the CPU oracle."""
    import bench
    from diffusion_nlc_amd import script_util
    from diffusion_nlc_amd.experiments import EDMImageExperiment
    from diffusion_nlc_amd.filler import fill_state_dict
    from oracle import edm
    from oracle.loop import EdmOracle
    mc = dict(img_resolution=32, in_channels=3, out_channels=3, augment_dim=9, model_channels=128, channel_mult=[2, 2, 2],
              num_blocks=4, attn_resolutions=[16], dropout=0.0, sigma_block=2, sigma_dropout=0.0)
    eps, sig, _ = script_util.create_edm_sigma_eps_model(**mc)
    tmpl = eps.state_dict()
    for k in tmpl:
        if k.endswith("resample_filter"):
            tmpl[k] = torch.ones_like(tmpl[k]) / 4.0
    sd_e = fill_state_dict(tmpl, seed=0)
    sd_s = fill_state_dict(sig.state_dict(), seed=1, overrides={"final_mlp.weight": 0.1, "final_mlp.bias": 0.5})
    eps.load_state_dict(sd_e); sig.load_state_dict(sd_s)
    eps.to("cuda:0"); sig.to("cuda:0")
    bench.set_precision(eps, bench.PRECISIONS[prec])
    bench.set_precision(sig, bench.PRECISIONS[prec])
    B, steps = 4, 6
    cfg = edm.EdmConfig(img_resolution=32, in_channels=3, out_channels=3, augment_dim=9, model_channels=128, channel_mult=(2, 2, 2),
                        num_blocks=4, attn_resolutions=(16,), sigma_block=2)
    _, dim = edm.sigma_dims(cfg)
    o = EdmOracle(lambda x, t: edm.unet(sd_e, cfg, x, t, "forward"), lambda x, t: edm.unet(sd_e, cfg, x, t, "encode"),
                  lambda f: edm.sigma_net(sd_s, dim, cfg.sigma_block, f), (3, 32, 32), num_timesteps=steps, norm_min=0.0, norm_max=54.63)
    lat = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(77))
    x_cpu = o.edm_sampler(lat, style="pred_partial,pred", norm_eps="000", eps_ratio=0.5, eps_scale=1.0, use_second_order=True)
    exp = EDMImageExperiment(eps, None, batch_size=B, data_shape=(3, 32, 32), seed=0, device="cuda:0", num_timesteps=steps)
    exp.set_model(eps, sig, learn_epsvar=False)
    exp.set_norm_maxmin(0.0, 54.63)
    x_gpu = exp.edm_sampler(shape=(B, 3, 32, 32), latents=lat, style="pred_partial,pred", norm_eps="000", eps_ratio=0.5, eps_scale=1.0,
                            use_second_order=True)
    err = (x_gpu.cpu().double() - x_cpu.double()).abs().max().item()
    rms = (x_gpu.cpu().double() - x_cpu.double()).pow(2).mean().sqrt().item()
    print(f"EDM CIFAR-10 SongUNet {prec} / f64 state, 6-step Heun+NLC: HIP vs CPU oracle L-inf = {err:.3e}, RMS {rms:.3e} "
          f"(sample RMS {x_cpu.double().pow(2).mean().sqrt().item():.3e})")
    assert x_gpu.dtype == torch.float64 and (err if prec.startswith("f32") else rms) <= EDM_GATES[prec]
