"""nlc_sched_step against the reference's own pred_xprev / get_eps_logvar for ALL seven live sampler variants
(src/schedulers.py:432-627) x {fixedsmall, fixedlarge, learned} x eta in {0, 0.85}.

The goldens (tests/golden/sched.npz, ``px_<sampler>_<var>_<eta>`` and ``lv_<var>``) were recorded from the imported
reference with ``torch.randn_like`` replaced by the stored noise tensor (tests/golden/make_golden.py:163-187), i.e. the
host-ordered noise injection the HIP path uses.  Two routes into the kernel are checked:

  * ``Scheduler.pred_xprev`` (the reference-shaped method): x0 and the log-variance are handed in, phases = 2;
  * the route ``denoise_loop`` takes: the kernel derives the log-variance itself from (sigma_t, sigma_prev, the
    learned-variance channels of the network output) - get_eps_logvar fused into the update.

Tolerance 1e-6 relative to the tensor's scale (the algebra is compiled with -ffp-contract=off; what is left are the
ulp-level differences of expf / logf / sqrtf between the device and ATen's CPU vector math).
"""
import pytest
import torch

from tests.util import load_npz

pytestmark = pytest.mark.gpu

SAMPLERS = ("ddpm", "ddim", "ddim_simple", "ddim_orig", "ddim_simple_orig", "ddim_simple_drag", "ddpm_orig")
VARS = ("fixedsmall", "fixedlarge", "learned")
ETAS = (0.0, 0.85)
TOL = 1e-6


def _sampler(name, var, eta):
    from diffusion_nlc_amd.schedulers import get_sampler
    s = get_sampler(name, 1000, 50, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var=var, eta=eta)
    s.to("cuda:0")
    return s


def _rel(got, ref):
    return ((got.detach().cpu().double() - ref.double()).abs().max() / ref.double().abs().max().clamp(min=1e-12)).item()


@pytest.fixture(scope="module")
def g():
    return load_npz("sched")


@pytest.mark.parametrize("eta", ETAS)
@pytest.mark.parametrize("var", VARS)
@pytest.mark.parametrize("name", SAMPLERS)
def test_pred_xprev_method_matches_reference(g, name, var, eta):
    dev = "cuda:0"
    s = _sampler(name, var, eta)
    x0, xt, eps, noise, st, sp = (g[k].to(dev) for k in ("px_x0", "px_xt", "px_eps", "px_noise", "px_st", "px_sp"))
    lv = g[f"lv_{var}"].to(dev)
    xp = s.pred_xprev(x0=x0, eps=eps, sigma_t=st, sigma_prev=sp, xt=xt, log_variance=lv, noise=noise)
    torch.cuda.synchronize()
    ref = g[f"px_{name}_{var}_{eta}"]
    err = _rel(xp, ref)
    assert torch.isfinite(xp).all() and err <= TOL, (name, var, eta, err)


@pytest.mark.parametrize("eta", ETAS)
@pytest.mark.parametrize("var", VARS)
@pytest.mark.parametrize("name", SAMPLERS)
def test_sched_step_kernel_derives_logvar_itself(g, name, var, eta):
    """denoise_loop's route: network output [B][2C][H][W] (eps || learned-variance fraction), var_mode decides."""
    from diffusion_nlc_amd import ops
    from diffusion_nlc_amd._ext import SCHED_VARIANTS, VAR_MODES, SchedDesc
    dev = "cuda:0"
    s = _sampler(name, var, eta)
    x0, xt, eps, learned, noise = (g[k].to(dev).contiguous() for k in ("px_x0", "px_xt", "px_eps", "px_learned", "px_noise"))
    st, sp = g["px_st"].reshape(-1).to(dev).contiguous(), g["px_sp"].reshape(-1).to(dev).contiguous()
    B, C = x0.shape[0], x0.shape[1]
    net_out = torch.cat([eps, learned], dim=1).contiguous()
    x0c, xp = x0.clone(), torch.empty_like(x0)
    eps_used = torch.empty_like(x0)
    nan = torch.zeros(1, device=dev, dtype=torch.int32)
    d = SchedDesc(xt=xt.data_ptr(), eps_out=net_out.data_ptr(), noise=noise.data_ptr(), sigma_t=st.data_ptr(),
                  sigma_prev=sp.data_ptr(), x0=x0c.data_ptr(), x_prev=xp.data_ptr(), eps_used=eps_used.data_ptr(), B=B, C=C,
                  Cnet=2 * C, HW=x0.numel() // (B * C), variant=SCHED_VARIANTS[name], clip=0, var_mode=VAR_MODES[var],
                  phases=2, eta=float(s.eta), min_var_coef=float(s.min_var_coef))
    ops.sched_step(d, nan)
    torch.cuda.synchronize()
    ref = g[f"px_{name}_{var}_{eta}"]
    err = _rel(xp, ref)
    assert int(nan.item()) == 0 and err <= TOL, (name, var, eta, err)
    assert torch.equal(x0c, x0)                                  # phases = 2 leaves x0 alone
    # the eps the update used: the passed-in one, or the one recomputed from the given x0 for the *_orig / drag variants
    if name in ("ddim_orig", "ddim_simple_orig", "ddim_simple_drag"):
        want = (g["px_xt"] - g["px_x0"]) / g["px_st"]
    else:
        want = g["px_eps"]
    assert _rel(eps_used, want) <= TOL


@pytest.mark.parametrize("var", VARS)
def test_get_eps_logvar_matches_reference(g, var):
    s = _sampler("ddim", var, 0.0)
    lv = s.get_eps_logvar(g["px_st"].to("cuda:0"), g["px_sp"].to("cuda:0"), g["px_learned"].to("cuda:0") if var == "learned" else None)
    assert _rel(lv, g[f"lv_{var}"]) <= TOL


def test_stochastic_variant_without_noise_is_rejected(g):
    from diffusion_nlc_amd import ops
    from diffusion_nlc_amd._ext import NlcError, SCHED_VARIANTS, VAR_MODES, SchedDesc
    dev = "cuda:0"
    x0 = g["px_x0"].to(dev).contiguous()
    st = g["px_st"].reshape(-1).to(dev).contiguous()
    xp = torch.empty_like(x0)
    d = SchedDesc(xt=x0.data_ptr(), eps_out=x0.data_ptr(), sigma_t=st.data_ptr(), sigma_prev=st.data_ptr(), x0=x0.data_ptr(),
                  x_prev=xp.data_ptr(), B=3, C=3, Cnet=3, HW=64, variant=SCHED_VARIANTS["ddpm"], clip=0,
                  var_mode=VAR_MODES["fixedsmall"], phases=2, eta=1.0, min_var_coef=1e-4)
    with pytest.raises(NlcError):
        ops.sched_step(d)
