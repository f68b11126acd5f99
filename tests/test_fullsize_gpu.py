"""BASELINE configs 2, 3 and 4 at their FULL model sizes against the CPU oracle, in every precision bench.py can run
(bench.PRECISIONS): f32 (exact f32 MFMA) and f32x3 (f32 storage, split-f16 three-pass matrix math) carry the north-star's
1e-3 per-pixel gate (ADM-256: after each of the first ten timesteps of the headline 50-step schedule); bf16 (the benchmarked
dtype) and f16 (the reference's own use_fp16 mode) are gated by tolerances against the HIP f32x3 path that those tests pin to the
oracle: teacher-forced corrected sigma / eps per timestep, and the statistics of the free-running 50-timestep sample at B = 16.  Plus size-independent properties of the benchmarked path: determinism,
per-sample independence (batch-permutation equivariance - no op on the path couples samples), range."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_policy")]


def _set_precision(exp, name):
    import bench
    for m in (exp.model, exp.sigma_model):
        bench.set_precision(m, bench.PRECISIONS[name])


ADM_STEPS = 10          # timesteps of the headline 50-step schedule the full-size oracle is stepped through (~1.4 s each)


@pytest.fixture(scope="module")
def adm256():
    """The headline experiment: ADM-256, the 50-step DDIM+NLC schedule of BASELINE.json configs[1] (tests stop after a few
    timesteps with ``max_steps``)."""
    import bench
    dev = torch.device("cuda:0")
    return bench.make_experiment(dict(bench.ADM256), dev, bench.PRECISIONS["bf16"], 4, 50)


def _run(exp, xT, max_steps=2, **kw):
    x, logs = exp.denoise_loop(shape=tuple(xT.shape), xT=xT, style="pred", norm_eps=True, refine_prior_sigma=True,
                               return_log=kw.pop("return_log", False), chunk_size=1, sigma_pred_threshold=960, max_steps=max_steps, **kw)
    return x if not logs[3] else (x, logs)


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


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_adm256_skip_projection_with_normalised_side_output_equals_the_two_passes(adm256, prec):
    """ops.FUSE_GN_SKIP at the benchmark batch (the launches only qualify from ~1 500 output tiles on): one evaluation of the
    full-size network with the ResBlocks' skip projection writing act(GroupNorm(x)) as a side output (nlc_conv_desc.norm_out) against
    the separate GroupNorm pass + plain 1x1 - the skip outputs are bit-identical, the normalised activations differ by one rounding
    of a*x+b in a few elements, so eps agrees far inside the 16-bit noise of the network (~1e-2 of scale)."""
    from diffusion_nlc_amd import ops
    exp = adm256
    _set_precision(exp, prec)
    try:
        g = torch.Generator().manual_seed(11)
        x = (torch.randn(16, 3, 256, 256, generator=g) * 20).to("cuda:0")
        t = torch.linspace(900.0, 100.0, 16, device="cuda:0")
        c_in = torch.full((16,), 0.04, device="cuda:0")
        outs = {}
        for flag in (True, False, True):
            old = ops.FUSE_GN_SKIP
            ops.FUSE_GN_SKIP = flag
            try:
                outs.setdefault(flag, []).append(exp.model.run(x, t, mode="forward", in_scale=c_in).clone())
            finally:
                ops.FUSE_GN_SKIP = old
        fused, fused2, plain = outs[True][0], outs[True][1], outs[False][0]
        assert torch.equal(fused, fused2)
        scale = plain.float().pow(2).mean().sqrt().item()
        rms = (fused.float() - plain.float()).pow(2).mean().sqrt().item()
        assert torch.isfinite(fused).all() and rms <= 2e-3 * scale, (rms, scale)
    finally:
        _set_precision(exp, "bf16")


@pytest.fixture(scope="module")
def adm256_oracle():
    """The CPU oracle on the headline model itself (ADM-256, 614 M parameters, 256x256): the first ADM_STEPS timesteps of the
    headline 50-step DDIM+NLC schedule for one image (~1.4 s of oracle time per timestep on the GPU box's 16 host cores).
    Returns (xT, per-timestep trace: clipped x0, corrected sigma_t)."""
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
    s = oracle_sampler("ddim", 1000, 50, **kw)
    o = DiffusionOracle(lambda x, t: adm.unet(sd_e, ucfg, x, t, "forward"), lambda x, t: adm.unet(sd_e, ucfg, x, t, "encode"),
                        lambda f: adm.sigma_net(sd_s, scfg, f), s, (3, 256, 256), learn_epsvar=True, norm_min=0.0,
                        norm_max=440.0, clip_fn="dynamic")
    z = torch.randn((1, 3, 256, 256), generator=torch.Generator().manual_seed(99))
    xT = z / (1 / (s.sampling_sigmas[0] ** 2 + 1)).sqrt()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    trace = {"x0": [], "sigma_t": []}
    xt = xT
    with torch.no_grad():
        for n in range(ADM_STEPS):                                          # ExperimentDiffusion.denoise_loop, src/experiments.py:346-381
            eps, lv, st, sp = o.get_denoise_vector(xt, s.timesteps[n], s.sampling_sigmas[n], s.sampling_sigmas[n + 1], "pred", True, True)
            x0 = o.clip(s.pred_xstart(xt, eps, st))
            trace["x0"].append(x0.clone()); trace["sigma_t"].append(st.reshape(-1).clone())
            xt = s.pred_xprev(x0=x0, eps=eps, sigma_t=st, sigma_prev=sp, xt=xt, log_variance=lv)
    return xT, trace


@pytest.mark.parametrize("prec", ["f32", "f32x3"])
def test_adm256_f32_ten_steps_match_the_oracle_at_full_size(adm256, adm256_oracle, prec):
    """The headline model on the headline schedule against the CPU oracle: the first TEN of the 50 DDIM+NLC timesteps (refine ->
    encode -> sigma net -> corrected sigma / t -> eps forward -> learned variance, dynamic-threshold clip -> scheduler update)
    for one image; per-pixel L-inf of the clipped x0 <= 1e-3 (the north-star tolerance) after EVERY one of them, corrected
    sigma within 1e-5 relative - in exact f32 and in the split-f16 matrix mode (f32 storage, three 16-bit MFMA passes per
    product: the path that carries the gate at a usable speed).  Runs under the production dispatch and with the halo kernel
    forced.  (All 50 timesteps, with the float64 leg: tools/parity_trace.py, profiles/r04_parity_trace_f64.json.)"""
    exp = adm256
    xT, trace = adm256_oracle
    _set_precision(exp, prec)
    try:
        x_gpu, logs = _run(exp, xT, max_steps=ADM_STEPS, return_log=True)
        sig = [s.double() for s in exp.sigma_trace]
    finally:
        _set_precision(exp, "bf16")
    errs = [(logs[3][i].double() - trace["x0"][i].double()).abs().max().item() for i in range(ADM_STEPS)]
    srel = [((sig[i] - trace["sigma_t"][i].double()).abs() / trace["sigma_t"][i].double()).max().item() for i in range(ADM_STEPS)]
    print(f"ADM-256 {prec}, {ADM_STEPS} of 50 DDIM+NLC timesteps, 1 image: HIP vs CPU oracle x0 L-inf per timestep "
          + " ".join(f"{e:.1e}" for e in errs) + f"; corrected sigma rel max {max(srel):.1e}")
    assert torch.equal(x_gpu, logs[3][-1])                                # the loop returns the last clipped x0
    assert max(errs) <= 1e-3 and max(srel) <= 1e-5, (errs, srel)


# ---- the 16-bit precisions (bf16 = the benchmarked dtype, f16 = the reference's own use_fp16 mode) -----------------------------------
# A 16-bit trajectory of this random-weight network is not pixel-comparable with the f32 one beyond a few timesteps (DESIGN.md §2:
# sigma_0 = 100 turns a relative error d of the corrected sigma into ~100 d |eps| of x0, and the sampling map expands errors ~x1.13
# per timestep), so their gates are TOLERANCES on quantities that stay comparable, measured against the HIP f32x3 path - which the
# test above pins to the CPU oracle on the same schedule:
#   (1) teacher-forced, per timestep: from the f32x3 run's own state x_t at timesteps TF_STEPS of the 50, one 16-bit NLC step: the
#       corrected sigma (what NLC is for) within SIGMA_TOL relative for every image; and one 16-bit evaluation of the eps network at
#       the SCHEDULED (sigma, t) - the corrected sigma picks t through a discontinuous table lookup, so eps at the corrected pair
#       compares two different timesteps whenever the 16-bit sigma lands in the next bin - within EPS_TOL relative RMS;
#   (2) free-running, B = 16, same x_T, after 5 / 10 / 20 / 50 timesteps: the 16-bit x0 of EVERY image closer (RMS) to the f32x3 x0
#       of the same seed than any two f32x3 x0 of different seeds are to each other, for 5, 10 and 20 timesteps (measured on
#       MI355X: bf16 0.31 / 0.15 / 0.14 against 0.76 / 0.24 / 0.16; f16 median 2e-3 / 2.6e-2 / 7e-2); after all 50 the
#       trajectories have decorrelated (same-seed RMS 0.13-0.14 = the x0 std: a different, equally valid sample) and the gate is
#       on the population: the median same-seed RMS still below the closest pair of seeds, and - at every horizon - the batch
#       mean within POP_MEAN_TOL and the batch-mean per-image standard deviation within POP_STD_TOL of the f32x3 run's.
TF_STEPS = (0, 1, 4, 9, 19, 29, 39, 49)
SIGMA_TOL = {"bf16": 2e-2, "f16": 2.5e-3}        # 2^-8 vs 2^-11 significand: a factor 8 between them
EPS_TOL = {"bf16": 2e-2, "f16": 3e-3}
HORIZONS = (5, 10, 20, 50)
POP_MEAN_TOL = 1e-2          # |batch mean of the 16-bit x0 - batch mean of the f32x3 x0|, x0 in [-1, 1] (measured <= 6.8e-3)
POP_STD_TOL = 5e-2           # |batch mean of the per-image std, ratio - 1| (measured <= 2.8e-2)


@pytest.fixture(scope="module")
def adm256_f32x3_run(adm256):
    """The pinned path's own run: f32x3, B = 16, all 50 timesteps, logged (x_t and corrected sigma of every timestep).
    Production dispatch (the conv_policy fixture is function-scoped; this runs under whatever the first user has set, and the
    f32 / f32x3 paths take no split-K either way)."""
    exp = adm256
    g = torch.Generator().manual_seed(4321)
    sigma0 = exp.scheduler.sampling_sigmas[0]
    xT = torch.randn(16, 3, 256, 256, generator=g) / (1 / (sigma0 ** 2 + 1)).sqrt()
    _set_precision(exp, "f32x3")
    try:
        x, logs = _run(exp, xT, max_steps=None, return_log=True)
        eps_tf = {}
        S = exp.scheduler
        for k in TF_STEPS:
            args = (exp.xt_trace[k], S.timesteps[k].item(), S.sampling_sigmas[k].item(), S.sampling_sigmas[k + 1].item())
            _, _, st, _ = exp.get_denoise_vector(*args, "pred", True, True)             # NLC step: the corrected sigma
            e, _, _, _ = exp.get_denoise_vector(*args, "base", True, False)             # eps at the SCHEDULED (sigma, t): the network alone
            eps_tf[k] = (e.cpu(), st.view(-1).cpu())
        return dict(xT=xT, x=x, x0=[t.clone() for t in logs[3]], xt=[t.clone() for t in exp.xt_trace],
                    sigma=[s.clone() for s in exp.sigma_trace], eps_tf=eps_tf)
    finally:
        _set_precision(exp, "bf16")


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_adm256_16bit_teacher_forced_sigma_and_eps_vs_f32x3(adm256, adm256_f32x3_run, prec):
    exp, ref = adm256, adm256_f32x3_run
    S = exp.scheduler
    _set_precision(exp, prec)
    worst_s, worst_e = 0.0, 0.0
    try:
        for k in TF_STEPS:
            args = (ref["xt"][k], S.timesteps[k].item(), S.sampling_sigmas[k].item(), S.sampling_sigmas[k + 1].item())
            _, _, st, _ = exp.get_denoise_vector(*args, "pred", True, True)
            e, _, _, _ = exp.get_denoise_vector(*args, "base", True, False)
            e_ref, s_ref = ref["eps_tf"][k]
            assert torch.allclose(s_ref, ref["sigma"][k], rtol=1e-6, atol=0)          # the teacher-forced f32x3 step reproduces its run
            srel = ((st.view(-1).cpu().double() - s_ref.double()).abs() / s_ref.double()).max().item()
            d = (e.cpu().double() - e_ref.double()).flatten(1)
            erel = (d.pow(2).mean(1).sqrt() / e_ref.double().flatten(1).pow(2).mean(1).sqrt()).max().item()
            print(f"ADM-256 {prec} timestep {k + 1}/50 from the f32x3 state, B=16: corrected sigma rel (max over images) {srel:.2e}, "
                  f"eps relative RMS (max over images) {erel:.2e}")
            worst_s, worst_e = max(worst_s, srel), max(worst_e, erel)
    finally:
        _set_precision(exp, "bf16")
    assert worst_s <= SIGMA_TOL[prec] and worst_e <= EPS_TOL[prec], (worst_s, worst_e)


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_adm256_16bit_sample_statistics_vs_f32x3(adm256, adm256_f32x3_run, prec):
    exp, ref = adm256, adm256_f32x3_run
    rows = {}
    _set_precision(exp, prec)
    try:
        for n in HORIZONS:
            rows[n] = _run(exp, ref["xT"], max_steps=n)
    finally:
        _set_precision(exp, "bf16")
    for n in HORIZONS:
        a, b = rows[n].double().flatten(1), ref["x0"][n - 1].double().flatten(1)
        same = (a - b).pow(2).mean(1).sqrt()                             # [16] RMS to the f32x3 x0 of the same seed
        cross = torch.cdist(b, b) / b.shape[1] ** 0.5                     # RMS between f32x3 x0 of different seeds
        cross_min = (cross + torch.eye(16, dtype=cross.dtype) * 1e9).min().item()
        dmean = (a.mean(1) - b.mean(1)).abs().max().item()
        pop_mean = abs(a.mean().item() - b.mean().item())
        pop_std = abs(a.std(1).mean().item() / b.std(1).mean().item() - 1)
        print(f"ADM-256 {prec} vs f32x3 after {n} timesteps, B=16: RMS to the same-seed f32x3 x0 max {same.max().item():.3e} / median "
              f"{same.median().item():.3e}; min RMS between two f32x3 seeds {cross_min:.3e}; per-image mean diff max {dmean:.2e}; batch mean diff "
              f"{pop_mean:.2e}, batch-mean std ratio - 1 {pop_std:.2e} (x0 std {b.std(1).mean().item():.3e})")
        assert torch.isfinite(rows[n]).all() and rows[n].abs().max() <= 1.0 + 1e-6
        assert (same.max().item() if n <= 20 else same.median().item()) <= cross_min, (n, same.max().item(), same.median().item(), cross_min)
        assert pop_mean <= POP_MEAN_TOL and pop_std <= POP_STD_TOL, (n, pop_mean, pop_std)


# cfg 4 / cfg 3 gates.  f32 and f32x3: the north-star's 1e-3 per-pixel L-inf on the final sample against the CPU oracle.  The 16-bit
# types get the same kind of TOLERANCES as ADM-256 above, against the HIP f32x3 path of the same test (which the f32x3 parametrisation
# pins to the oracle): teacher-forced from the f32x3 run's own states, the NLC-corrected sigma within SIGMA_TOL and the eps network at
# the scheduled (sigma, t) within EPS_TOL relative RMS (cfg 4: every timestep of the run; cfg 3: four noise levels spanning the
# schedule); the free-running 16-bit sample is reported (at sigma_0 = 100 the first x0 = xt - 100 eps turns a 1 % error of eps into an
# O(1) error of x0 and the +-1 clamp saturates single pixels: per-pixel distances of a 16-bit trajectory say nothing there) and
# gated only relative to the sample's own scale for cfg 3, whose six Heun steps stay correlated: RMS <= EDM_RMS_TOL x sample RMS.
EDM_RMS_TOL = {"bf16": 1e-2, "f16": 2e-3}        # measured (round 3): 4.4e-3 and 5.6e-4 of the sample RMS


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
    assert known == 0.0 and torch.isfinite(x_gpu).all()
    if prec.startswith("f32"):
        assert err <= 1e-3
        return
    # 16-bit: teacher-forced against the f32x3 path at every timestep of an f32x3 run (see the comment above the ADM-256 tests)
    bench.set_precision(eps, bench.PRECISIONS["f32x3"]); bench.set_precision(sig, bench.PRECISIONS["f32x3"])
    exp.denoise_loop(shape=(B, 3, res, res), xT=xT, style="pred", constrain_fn=cf.bind(y, (B, 3, res, res)), norm_eps=True,
                     refine_prior_sigma=True, return_log=True, chunk_size=1, sigma_pred_threshold=960)
    xts = [t.clone() for t in exp.xt_trace]

    def one(k):
        args = (xts[k], s.timesteps[k].item(), s.sampling_sigmas[k].item(), s.sampling_sigmas[k + 1].item())
        _, _, st, _ = exp.get_denoise_vector(*args, "pred", True, True)
        e, _, _, _ = exp.get_denoise_vector(*args, "base", True, False)
        return e.cpu().double(), st.view(-1).cpu().double()

    refs = [one(k) for k in range(steps)]
    bench.set_precision(eps, bench.PRECISIONS[prec]); bench.set_precision(sig, bench.PRECISIONS[prec])
    for k in range(steps):
        e, st = one(k)
        srel_k = ((st - refs[k][1]).abs() / refs[k][1]).max().item()
        erel_k = ((e - refs[k][0]).pow(2).mean().sqrt() / refs[k][0].pow(2).mean().sqrt()).item()
        print(f"  {prec} timestep {k + 1}/{steps} from the f32x3 state: corrected sigma rel {srel_k:.2e}, eps relative RMS {erel_k:.2e}")
        assert srel_k <= SIGMA_TOL[prec] and erel_k <= EPS_TOL[prec], (k, srel_k, erel_k)


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
    srms = x_cpu.double().pow(2).mean().sqrt().item()
    print(f"EDM CIFAR-10 SongUNet {prec} / f64 state, 6-step Heun+NLC: HIP vs CPU oracle L-inf = {err:.3e}, RMS {rms:.3e} (sample RMS {srms:.3e})")
    assert x_gpu.dtype == torch.float64 and torch.isfinite(x_gpu).all()
    if prec.startswith("f32"):
        assert err <= 1e-3
        return
    assert rms <= EDM_RMS_TOL[prec] * srms
    # teacher-forced at four noise levels spanning the schedule, against the f32x3 path on the same states
    g = torch.Generator().manual_seed(78)
    sig_levels = (80.0, 9.0, 1.0, 0.05)
    xs = [(torch.randn(B, 3, 32, 32, generator=g, dtype=torch.float64) * (sg ** 2 + 0.25) ** 0.5).to("cuda:0") for sg in sig_levels]

    def one(k):
        sg = torch.tensor(sig_levels[k], dtype=torch.float64)
        _, _, st, _ = exp.get_denoise_vector(xs[k], sg, sg * 0.5, style="pred", norm_eps=False, refine_prior_sigma=False)
        e, _, _, _ = exp.get_denoise_vector(xs[k], sg, sg * 0.5, style="base", norm_eps=False, refine_prior_sigma=False)
        return e.cpu().double(), st.reshape(-1).cpu().double()

    bench.set_precision(eps, bench.PRECISIONS["f32x3"]); bench.set_precision(sig, bench.PRECISIONS["f32x3"])
    refs = [one(k) for k in range(len(sig_levels))]
    bench.set_precision(eps, bench.PRECISIONS[prec]); bench.set_precision(sig, bench.PRECISIONS[prec])
    for k, sg in enumerate(sig_levels):
        e, st = one(k)
        srel_k = ((st - refs[k][1]).abs() / refs[k][1].abs()).max().item()
        erel_k = ((e - refs[k][0]).flatten(1).pow(2).mean(1).sqrt() / refs[k][0].flatten(1).pow(2).mean(1).sqrt()).max().item()
        print(f"  {prec} at sigma {sg}: corrected sigma rel {srel_k:.2e}, eps relative RMS (max over images) {erel_k:.2e}")
        assert srel_k <= SIGMA_TOL[prec] and erel_k <= EPS_TOL[prec], (sg, srel_k, erel_k)
