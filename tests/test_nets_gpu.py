"""Whole-network parity on the GPU: HIP networks (through the C ABI) vs the CPU oracle and the
golden outputs of the reference, on the tiny configurations of tests/golden/specs.json.

f32 paths (exact f32 MFMA and the split-f16 matrix mode): L-inf <= 1e-3 (north-star tolerance; observed ~1e-5).  16-bit paths:
reported against the same reference with a loose gate (bf16 5e-2, f16 8e-3 of the output scale) - 8 / 11 significand bits
cannot meet 1e-3.
"""
import pytest
import torch

from tests.test_host_cpu import build_product
from tests.util import load_npz, max_err, oracle_nets, state_dicts

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("conv_policy")]
TAGS = ["adm_tiny", "adm_tiny_b", "simple_tiny", "edm_tiny"]


def _models(tag, dtype, matmul="native"):
    eps, sig, _ = build_product(tag)
    e, s = state_dicts(tag)
    eps.load_state_dict(e)
    sig.load_state_dict(s)
    eps.to("cuda:0").set_compute_dtype(dtype).set_matmul(matmul)
    sig.to("cuda:0").set_compute_dtype(dtype).set_matmul(matmul)
    return eps, sig


@pytest.mark.parametrize("matmul", ["native", "f16x3"])
@pytest.mark.parametrize("tag", TAGS)
def test_f32_networks_match_reference(tag, matmul):
    """f32 storage; "native" = exact f32 MFMA, "f16x3" = split-f16 three-pass matrix math in the convolutions: both inside 1e-3."""
    g = load_npz(f"net_{tag}")
    eps, sig = _models(tag, torch.float32, matmul)
    out = eps(g["x"], g["t"]).cpu()
    feat = eps.encode(g["x"], g["t"]).cpu()
    r = sig(feat).cpu()
    assert out.shape == g["out"].shape and feat.shape == g["feat"].shape and r.shape == g["r"].shape
    e_out, e_feat, e_r = max_err(out, g["out"]), max_err(feat, g["feat"]), max_err(r, g["r"])
    print(f"{tag}: f32 ({matmul}) L-inf out {e_out:.2e} feat {e_feat:.2e} r {e_r:.2e}")
    assert e_out < 1e-3 and e_feat < 1e-3 and e_r < 1e-3
    if not tag.startswith("edm"):
        o2, f2 = eps.forward_and_encode(g["x"], g["t"])
        assert torch.equal(o2.cpu(), out) and torch.equal(f2.cpu(), feat)      # deterministic kernels


@pytest.mark.parametrize("dtype,gate", [(torch.bfloat16, 5e-2), (torch.float16, 8e-3)], ids=["bf16", "f16"])
@pytest.mark.parametrize("tag", TAGS)
def test_16bit_networks_track_reference(tag, dtype, gate):
    g = load_npz(f"net_{tag}")
    eps, sig = _models(tag, dtype)
    out = eps(g["x"], g["t"]).cpu()
    feat = eps.encode(g["x"], g["t"]).cpu()
    r = sig(feat).cpu()
    so, sf = g["out"].abs().max().item(), g["feat"].abs().max().item()
    e_out, e_feat, e_r = max_err(out, g["out"]), max_err(feat, g["feat"]), max_err(r, g["r"])
    print(f"{tag}: {dtype} L-inf out {e_out:.2e} (scale {so:.2f}) feat {e_feat:.2e} (scale {sf:.2f}) r {e_r:.2e}")
    assert e_out < gate * so and e_feat < gate * sf and e_r < gate


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_class_conditional_adm(dtype):
    """model(x, t, y) (src/unet_adm.py:479-480,652-654): golden from the reference's class-conditional UNetModel."""
    g = load_npz("net_adm_tiny_cc")
    eps, sig = _models("adm_tiny_cc", dtype)
    out = eps(g["x"], g["t"], g["y"]).cpu()
    feat = eps.encode(g["x"], g["t"], g["y"]).cpu()
    r = sig(feat).cpu()
    tol_o, tol_f = (1e-3, 1e-3) if dtype == torch.float32 else (5e-2 * g["out"].abs().max().item(), 5e-2 * g["feat"].abs().max().item())
    e_out, e_feat, e_r = max_err(out, g["out"]), max_err(feat, g["feat"]), max_err(r, g["r"])
    print(f"adm_tiny_cc: L-inf out {e_out:.2e} feat {e_feat:.2e} r {e_r:.2e}")
    assert e_out < tol_o and e_feat < tol_f and e_r < (1e-3 if dtype == torch.float32 else 5e-2)
    o2, f2 = eps.forward_and_encode(g["x"], g["t"], g["y"])
    assert torch.equal(o2.cpu(), out) and torch.equal(f2.cpu(), feat)
    with pytest.raises(AssertionError):
        eps(g["x"], g["t"])                                   # y is mandatory for a class-conditional model (:645-647)
    other = eps(g["x"], g["t"], torch.zeros_like(g["y"])).cpu()
    assert max_err(other, out) > 1e-4


def test_batch_independence():
    """Samples do not couple inside a batch (the property the multi-GPU sharding relies on)."""
    g = load_npz("net_adm_tiny")
    eps, sig = _models("adm_tiny", torch.float32)
    full = eps(g["x"], g["t"]).cpu()
    for b in range(2):
        one = eps(g["x"][b:b + 1], g["t"][b:b + 1]).cpu()
        assert torch.equal(one[0], full[b])


def test_sigma_training_batch_matches_the_reference_formulas():
    """SURVEY.md §8 f-4 (src/experiments.py:665-681): forward process, regression target and the microbatched frozen-encoder
    features of one sigma-net training iteration, against the reference's formulas on the CPU + the oracle's encoder."""
    from diffusion_nlc_amd.experiments import ImageExperiment
    from diffusion_nlc_amd.schedulers import get_sampler
    from oracle.sched import get_sampler as oracle_sampler
    eps, sig = _models("adm_tiny", torch.float32)
    s = get_sampler("ddim", 1000, 10, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall")
    s.to("cuda:0")
    exp = ImageExperiment(eps, s, batch_size=5, data_shape=(3, 64, 64), seed=3, device="cuda:0")
    exp.set_model(eps, sig, learn_epsvar=True)
    g = torch.Generator().manual_seed(17)
    x = torch.rand(5, 3, 64, 64, generator=g) * 2 - 1
    t = torch.tensor([0, 999, 500, 17, 640])
    noise = torch.randn(5, 3, 64, 64, generator=g) * 1.1
    feat, dist_real, noisy = exp.sigma_training_batch(x, t, noise, microbatch=2)          # microbatches of 2, 2, 1
    os_ = oracle_sampler("ddim", 1000, 10, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall")
    alpha = os_.alphas_cumprod[t].view(-1, 1, 1, 1)
    noisy_ref = x * alpha.sqrt() + noise * (1 - alpha).sqrt()                              # src/schedulers.py:327-328
    assert torch.equal(noisy.cpu(), noisy_ref)                                             # bit-exact f32 algebra
    dref = torch.linalg.vector_norm(noise, dim=(1, 2, 3), keepdim=True) / (3 * 64 * 64) ** 0.5
    assert ((dist_real.cpu() - dref).abs() / dref).max() < 1e-6
    _, enc_fn, _, _ = oracle_nets("adm_tiny")
    with torch.no_grad():
        feat_ref = enc_fn(noisy_ref, t.float())
    assert feat.shape == feat_ref.shape and max_err(feat.cpu(), feat_ref) < 1e-3


def test_adm_network_with_groupnorm_in_the_conv_prologue():
    """ops.FUSE_GN_CONV (off by default: measured slower, see ops.py): the whole tiny ADM forward with every eligible
    GroupNorm+SiLU(+FiLM) applied inside the consuming convolution against the default separate-pass build."""
    from diffusion_nlc_amd import ops
    g = load_npz("net_adm_tiny")
    eps, _ = _models("adm_tiny", torch.bfloat16)
    was_f, was_p = ops.FUSE_GN_CONV, ops.CONV_POLICY
    try:
        ops.CONV_POLICY = "halo"                 # the tiny maps reach the halo kernel (and with it the prologue) only when forced
        ops.FUSE_GN_CONV = False
        ref = eps(g["x"], g["t"]).cpu()
        ops.FUSE_GN_CONV = True
        got = eps(g["x"], g["t"]).cpu()
    finally:
        ops.FUSE_GN_CONV, ops.CONV_POLICY = was_f, was_p
    scale = ref.abs().max().item()
    assert max_err(got, ref) <= 3e-2 * scale and max_err(got, g["out"]) <= 5e-2 * g["out"].abs().max().item()


def test_graph_outputs_own_their_ride_along_statistics():
    """A feature map returned by a captured evaluation keeps valid GroupNorm totals after OTHER evaluations of the same module have
    re-zeroed the module's statistics arena (hipnet._own_output_stats): encode (graph) -> forward (graph) -> forward (eager) -> the
    sigma net on the kept feature map gives the bits of the immediate encode -> sigma order."""
    from diffusion_nlc_amd import ops
    g = load_npz("net_adm_tiny")
    eps, sig = _models("adm_tiny", torch.bfloat16)
    x, t = g["x"].to("cuda:0"), g["t"].to("cuda:0").float()
    feat0 = eps.run(x, t, mode="encode", feat_nhwc=True)
    r_ref = sig.run_nhwc(feat0).cpu()
    eps.use_graphs = True
    try:
        feat = eps.run(x, t, mode="encode", feat_nhwc=True)
        had = ops.ride_stats(feat)
        kept = None if had is None else had.clone()
        eps.run(x, t, mode="forward")                    # another graph of the same module: re-zeroes and rewrites the arena
        eps.use_graphs = False
        eps.run(x, t, mode="forward")                    # and an eager evaluation
        if kept is not None:
            assert torch.equal(ops.ride_stats(feat), kept)
        assert torch.equal(feat, feat0)
        r = sig.run_nhwc(feat).cpu()
    finally:
        eps.use_graphs = False
        eps.drop_graphs()
    assert torch.equal(r, r_ref)


def test_attention_base_flip_drops_captured_graphs():
    """ops.ATTN_BASE2 is baked into the packed q rows: a flip re-packs AND forgets the graphs that point into the old weights."""
    from diffusion_nlc_amd import ops
    g = load_npz("net_adm_tiny")
    eps, _ = _models("adm_tiny", torch.bfloat16)
    x, t = g["x"].to("cuda:0"), g["t"].to("cuda:0").float()
    was = ops.ATTN_BASE2
    eps.use_graphs = True
    try:
        a = eps.run(x, t, mode="forward").clone()
        ops.ATTN_BASE2 = not was
        eps.plan()
        assert not eps.__dict__.get("_graphs")
        b = eps.run(x, t, mode="forward").clone()
        ops.ATTN_BASE2 = was
        c = eps.run(x, t, mode="forward").clone()
    finally:
        ops.ATTN_BASE2 = was
        eps.use_graphs = False
        eps.drop_graphs()
    assert torch.equal(a, c)
    assert max_err(a.float().cpu(), b.float().cpu()) <= 5e-2 * a.float().abs().max().item()
