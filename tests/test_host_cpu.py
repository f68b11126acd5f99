"""CPU-side checks of the host logic: parameter specs vs the reference modules' state_dicts,
the C-ABI library's exports, and loud failure without a GPU."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

from tests.util import load_specs, state_dicts

ROOT = Path(__file__).resolve().parent.parent


def build_product(tag):
    import argparse
    from diffusion_nlc_amd import script_util
    cfgs = load_specs()["_configs"]
    c = dict(cfgs[tag])
    if tag.startswith("adm"):
        return script_util.create_sigma_eps_model(**c)
    if tag.startswith("simple"):
        ns = argparse.Namespace
        cfg = ns(model=ns(**{k: c[k] for k in ("ch", "out_ch", "ch_mult", "num_res_blocks", "attn_resolutions", "dropout",
                                                 "in_channels", "resamp_with_conv", "feat_layer", "type", "sigma_block",
                                                 "sigma_dropout")}),
                 data=ns(image_size=c["image_size"]), diffusion=ns(num_diffusion_timesteps=c["num_diffusion_timesteps"]))
        return script_util.create_simple_sigma_eps_model(cfg)
    return script_util.create_edm_sigma_eps_model(**c)


@pytest.mark.parametrize("tag", ["adm_tiny", "adm_tiny_b", "adm_tiny_cc", "simple_tiny", "edm_tiny"])
def test_param_spec_matches_reference_state_dict(tag):
    """Key names, order, shapes and dtypes equal what the reference modules produced (tests/golden/specs.json)."""
    specs = load_specs()
    eps, sig, fshape = build_product(tag)
    assert list(fshape) == specs[tag]["feat_shape"]
    for mod, ref in ((eps, specs[tag]["eps"]), (sig, specs[tag]["sigma"])):
        got = {k: [list(s), str(d).replace("torch.", "")] for k, (s, d) in mod.param_spec().items()}
        assert list(got.keys()) == list(ref.keys())
        assert got == ref
    e, s = state_dicts(tag)
    eps.load_state_dict(e)
    sig.load_state_dict(s)
    with pytest.raises(RuntimeError):
        eps.load_state_dict({k: v for k, v in list(e.items())[:-1]})


def test_library_exports_every_declared_symbol():
    from diffusion_nlc_amd import _ext
    header = (ROOT / "include" / "nlc_hip.h").read_text()
    declared = set(re.findall(r"\b(nlc_[A-Za-z0-9_]+)\s*\(", header))
    declared -= {"nlc_conv_desc", "nlc_sched_desc"}
    lib = _ext.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libnlc_hip.so does not export {name}"
        assert name in _ext.SIGNATURES, f"{name} is declared in nlc_hip.h but not bound in _ext.py"
    assert set(_ext.SIGNATURES) <= declared
    assert lib.nlc_version() == _ext.ABI_VERSION
    a, b = _ext.pack_dims(_ext.NLC_BF16)
    assert a % 16 == 0 and b % 8 == 0


def test_argument_validation_without_gpu():
    """Bad descriptors are rejected on the host before any launch."""
    from diffusion_nlc_amd import _ext
    lib = _ext.load()
    d = _ext.ConvDesc()
    assert lib.nlc_conv2d(ctypes.byref(d), 7, None) == -1
    assert b"dtype" in lib.nlc_last_error()
    assert lib.nlc_attention(None, None, 1, 1, 1, 64, 0, 0, None) == -1
    assert lib.nlc_conv_first(None, None, None, None, None, 1, 3, 8, 8, 16, 3, 3, 0, None, 0, 5, None) == -1        # stats_granule 5
    assert b"granule" in lib.nlc_last_error()
    d2 = _ext.ConvDesc(math=1)
    assert lib.nlc_conv2d(ctypes.byref(d2), _ext.NLC_BF16, None) == -1 and b"math" in lib.nlc_last_error()        # F16X3 is a mode of f32 tensors
    assert lib.nlc_groupnorm(None, None, 8, 0, 1, 1, 3, 1e-5, None, None, None, None, 0, 0, None, None, 0, None) == -1


def test_no_cpu_fallback():
    from diffusion_nlc_amd._ext import NlcError
    eps, sig, _ = build_product("adm_tiny_b")
    with pytest.raises(NlcError):
        eps(torch.zeros(1, 3, 32, 32), torch.zeros(1))
    with pytest.raises(NlcError):
        sig(torch.zeros(1, 64, 8, 8))


def test_continuous_t_and_redesigned_schedules_match_reference():
    """Host-side schedule construction for SURVEY §8 f-2 (continuous t by Interp1d; the sigma-redesign tail)."""
    import json
    import numpy as np
    from diffusion_nlc_amd.schedulers import get_sampler, redesign_sigma
    from tests.util import load_npz
    g = load_npz("cont_linear")
    s = get_sampler("ddim", 1000, 10, sigma_style="Linear", start_sigma=100, end_sigma=0.01, sampler_var="fixedsmall", eta=0.0,
                    continuous_t=True)
    assert s.continuous_t and s.timesteps.dtype == g["timesteps"].dtype
    assert torch.equal(s.timesteps, g["timesteps"]) and torch.equal(s.sampling_sigmas, g["sampling_sigmas"])
    assert s.device_t_slopes("cpu").shape == (999,)
    g = load_npz("proj_redesign")
    c = g["cfg"]
    s = get_sampler("ddim", 1000, c["num_timesteps"], sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall", eta=0.0)
    assert s.device_t_slopes("cpu") is None
    redesign_sigma(s, c["num_timesteps"], c["max_T"], c["cycle_size"], c["min_sigma"], c["max_sigma"], c["sigma_gamma"])
    assert s.continuous_t and s.sampling_sigmas.dtype == g["sampling_sigmas"].dtype == torch.float64
    assert torch.equal(s.sampling_sigmas, g["sampling_sigmas"])
    assert torch.equal(s.timesteps, g["timesteps"])


def test_multiply_shift_division_constants_are_exact():
    """conv_params.h: FastDiv (host side fills mul / shr per launch, the kernels compute q = mulhi(n, mul) >> shr).  Restated here:
    l = ceil(log2 d), mul = ceil(2^(31 + l) / d) < 2^32, q = (n * mul >> 32) >> (l - 1) must equal n // d for every 31-bit n
    (Granlund-Montgomery, N = 31); d = 1 is the kernels' special case."""
    import random
    rnd = random.Random(7)
    ds = list(range(2, 300)) + [2 ** k for k in range(1, 31)] + [2 ** k + 1 for k in range(1, 30)] + [2 ** k - 1 for k in range(2, 31)] + \
         [rnd.randrange(2, 2 ** 31) for _ in range(500)] + [65536 * 16, 256 * 256, 1023, 1025]
    for d in ds:
        l = (d - 1).bit_length()
        mul = ((1 << (31 + l)) + d - 1) // d
        assert mul < (1 << 32), d
        shr = l - 1
        ns = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 31) - 1, (1 << 31) - d, ((1 << 31) - 1) // d * d, ((1 << 31) - 1) // d * d - 1] + \
             [rnd.randrange(0, 1 << 31) for _ in range(64)]
        for n in ns:
            if 0 <= n < (1 << 31):
                assert ((n * mul) >> 32) >> shr == n // d, (n, d)


# ---- command lines of the two entry points vs the reference's own parsers (tests/golden/cli_flags.json) -------------------
# The ONLY tolerated difference: the default of --device.  The reference hard-codes its authors' card index (cuda:1 / cuda:5,
# image_sample.py:80, edm_image_sample.py:36), which does not exist on a one-GPU box; both entry points default to cuda:0.
CLI_DEFAULT_WAIVERS = {"--device"}


def _cli_flags(parser):
    return {a.option_strings[0]: dict(default=a.default, type=getattr(a.type, "__name__", None),
                                      choices=list(a.choices) if a.choices is not None else None)
            for a in parser._actions if a.option_strings and a.dest != "help"}


@pytest.mark.parametrize("cli", ["image_sample", "edm_image_sample"])
def test_cli_flags_match_reference(cli):
    """Every flag of the reference's parser exists here with the same default, type and choices; flags that exist only here are
    the documented extensions."""
    import importlib
    import json
    import math
    ref = json.loads((ROOT / "tests" / "golden" / "cli_flags.json").read_text())[cli]
    mod = importlib.import_module(cli)
    got = _cli_flags(mod.build_parser())
    missing = sorted(set(ref) - set(got))
    assert not missing, f"{cli}.py lacks reference flags {missing}"
    for flag, r in ref.items():
        g = got[flag]
        assert g["type"] == r["type"], (flag, g, r)
        assert g["choices"] == r["choices"], (flag, g, r)
        if flag in CLI_DEFAULT_WAIVERS:
            continue
        assert g["default"] == r["default"] and type(g["default"]) is type(r["default"]), (flag, g, r)
    extensions = {"image_sample": {"--synthetic", "--return_log", "--save_png", "--precision"},
                  "edm_image_sample": {"--synthetic", "--dtype", "--rho", "--S_churn", "--S_min", "--S_max", "--S_noise", "--save_png"}}[cli]
    assert set(got) - set(ref) == extensions
    for flag in extensions:                      # an extension left at its default must not change the reference's behaviour
        d = got[flag]["default"]
        assert d in (None, "", 0, 1, 7, "f32") or (isinstance(d, float) and math.isinf(d)), (flag, d)


@pytest.mark.parametrize("cfg", ["cifar10", "ffhq"])
def test_edm_get_args_matches_reference(cfg, tmp_path, monkeypatch):
    """edm_image_sample.get_args() on the reference's own invocation line, in a scratch tree with the two files it reads, returns
    the namespace the reference's get_args() returned (recorded by make_golden.py): derived paths, the values taken from the
    training run's args.json, get_default's per-config norm bounds (ffhq: norm_max 102.0 + its checkpoint / FID paths)."""
    import json
    import edm_image_sample
    fx = json.loads((ROOT / "tests" / "golden" / "cli_flags.json").read_text())["edm_get_args"]
    (tmp_path / "results" / cfg / "6").mkdir(parents=True)
    (tmp_path / "store" / "config").mkdir(parents=True)
    (tmp_path / "results" / cfg / "6" / "args.json").write_text(json.dumps(fx["saved_args_json"]))
    (tmp_path / "store" / "config" / f"{cfg}.yml").write_text("model:\n  img_resolution: 32\ndata:\n  channels: 3\n  image_size: 32\n")
    monkeypatch.chdir(tmp_path)
    args, config = edm_image_sample.get_args(["--config", cfg] + fx["argv"])
    want = fx["result"][cfg]
    got = vars(args)
    for k, v in want["args"].items():
        if k == "device":
            continue
        assert k in got and got[k] == v and type(got[k]) is type(v), (k, got.get(k), v)
    assert vars(config.model) == want["model"]
    assert (config.data.channels, config.data.image_size) == (3, 32)


def test_product_schedule_tables_match_reference():
    """diffusion_nlc_amd.schedulers (host tables, what every GPU loop consumes) bit for bit against the reference's tables
    (tests/golden/sched.npz): the three DDIM ladders of the BASELINE configs, the other beta schedules, set_alpha_to_one = False,
    the sigma -> t lookup grid; and the "EDM" / "Scaled" / "Linear" ladders (no reference fixture) against the oracle's restatement
    of src/schedulers.py:227-284, which the same fixture file pins."""
    from diffusion_nlc_amd.schedulers import get_sampler
    from oracle.sched import get_sampler as oracle_sampler
    from tests.util import load_npz
    g = load_npz("sched")
    for steps in (10, 50, 100):
        s = get_sampler("ddim", 1000, steps, sigma_style="DDIM", start_sigma=100, end_sigma=0, sampler_var="fixedsmall")
        assert torch.equal(s.timesteps, g[f"timesteps_{steps}"]) and torch.equal(s.sampling_sigmas, g[f"sampling_sigmas_{steps}"])
        assert torch.equal(torch.as_tensor(s.min_var_coef), torch.as_tensor(g[f"min_var_coef_{steps}"]))
    assert torch.equal(s.sigmas, g["sigmas"]) and torch.equal(s.alphas_cumprod, g["alphas_cumprod"])
    assert torch.equal(s.get_t_from_sigma(g["t_grid_sigma"]), g["t_grid_t"])
    for sched in ("quadratic", "cosine", "sigmoid"):
        s2 = get_sampler("ddim", 1000, 20, beta_schedule=sched, sigma_style="DDIM", start_sigma=0, end_sigma=0)
        assert torch.equal(s2.sigmas, g[f"sigmas_{sched}"]) and torch.equal(s2.timesteps, g[f"timesteps_{sched}"])
    s3 = get_sampler("ddim", 1000, 10, sigma_style="DDIM", start_sigma=100, end_sigma=0, set_alpha_to_one=False)
    assert torch.equal(s3.timesteps, g["timesteps_noalpha1"]) and torch.equal(s3.sampling_sigmas, g["sampling_sigmas_noalpha1"])
    for style, kw in (("EDM", {}), ("Scaled", dict(linear_scale=0.9)), ("Linear", {}), ("Scaled", dict(linear_scale=1.1, continuous_t=True)),
                      ("EDM", dict(continuous_t=True))):
        a = get_sampler("ddim", 1000, 12, sigma_style=style, start_sigma=80, end_sigma=0.02, sampler_var="fixedsmall", **kw)
        b = oracle_sampler("ddim", 1000, 12, sigma_style=style, start_sigma=80, end_sigma=0.02, sampler_var="fixedsmall", **kw)
        assert a.timesteps.dtype == b.timesteps.dtype and torch.equal(a.timesteps, b.timesteps), (style, kw)
        assert a.sampling_sigmas.dtype == b.sampling_sigmas.dtype and torch.equal(a.sampling_sigmas, b.sampling_sigmas), (style, kw)
        assert torch.equal(torch.as_tensor(a.min_var_coef), torch.as_tensor(b.min_var_coef)), (style, kw)
    with pytest.raises(ValueError):
        get_sampler("ddim", 1000, 12, sigma_style="Cosine", start_sigma=80, end_sigma=0.02)


# ---- NVIDIA EDM network pickles (torch_utils.persistence format) ------------------------------------------------------------------
def write_nvidia_format_pickle(path, state_dict):
    """A pickle in the FORMAT of NVIDIA's EDM checkpoints (/root/reference/torch_utils/persistence.py:123-131,185-208): every module
    is reduced to ``torch_utils.persistence._reconstruct_persistent_obj(dict(type='class', version=6, module_src=..., class_name=...,
    state=<the module's __dict__>))``.  Built from a state_dict: a tree of stand-in modules (EDMPrecond.model = the network; the
    network's ``enc`` / ``dec`` are real ``torch.nn.ModuleDict``s as upstream) whose leaves hold the tensors as parameters - or, for
    ``resample_filter``, as buffers.  ``module_src`` is a dummy string: nothing of the reference's source is written."""
    import pickle
    import sys
    import types
    import torch

    fake = types.ModuleType("torch_utils.persistence")

    def _reconstruct_persistent_obj(meta):                       # never called here: pickle stores it by qualified name
        raise RuntimeError("the stock hook would execute module_src")
    _reconstruct_persistent_obj.__module__ = "torch_utils.persistence"
    _reconstruct_persistent_obj.__qualname__ = "_reconstruct_persistent_obj"
    fake._reconstruct_persistent_obj = _reconstruct_persistent_obj
    pkg = types.ModuleType("torch_utils")
    pkg.persistence = fake

    class Node(torch.nn.Module):
        def __reduce__(self):
            meta = dict(type="class", version=6, module_src="# (source text elided)", class_name=type(self).__name__, state=dict(self.__dict__))
            return (_reconstruct_persistent_obj, (meta,), None)

    def build(prefix_items):
        node = Node()
        children = {}
        for key, t in prefix_items:
            head, _, rest = key.partition(".")
            if not rest:
                if key.endswith("resample_filter"):
                    node.register_buffer(key, t.clone())
                else:
                    node.register_parameter(key, torch.nn.Parameter(t.clone(), requires_grad=False))
            else:
                children.setdefault(head, []).append((rest, t))
        for head, items in children.items():
            if head in ("enc", "dec"):                          # upstream: torch.nn.ModuleDict of persistent blocks
                md = torch.nn.ModuleDict()
                blocks = {}
                for rest, t in items:
                    bname, _, brest = rest.partition(".")
                    blocks.setdefault(bname, []).append((brest, t))
                for bname, bitems in blocks.items():
                    md[bname] = build(bitems)
                node.add_module(head, md)
            else:
                node.add_module(head, build(items))
        return node

    precond = Node()
    precond.add_module("model", build(list(state_dict.items())))
    precond.sigma_data = 0.5
    old = {k: sys.modules.get(k) for k in ("torch_utils", "torch_utils.persistence")}
    sys.modules["torch_utils"], sys.modules["torch_utils.persistence"] = pkg, fake
    try:
        with open(path, "wb") as f:
            pickle.dump(dict(ema=precond, loss_fn=None, augment_pipe=None, dataset_kwargs=dict(resolution=32)), f)
    finally:
        for k, v in old.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_edm_pickle_reader_needs_no_dnnlib_and_executes_nothing(tmp_path):
    """diffusion_nlc_amd.edm_pickle: the state_dict of an NVIDIA-format network pickle, read with torch_utils / dnnlib absent, equals the
    tensors that went in, under the reference's key names; a pickle that references any other global is refused."""
    import pickle
    import sys
    import torch
    from diffusion_nlc_amd.edm_pickle import load_edm_pickle
    from diffusion_nlc_amd.filler import fill_state_dict
    from tests.util import load_specs, template_from_spec
    sd = fill_state_dict(template_from_spec(load_specs()["edm_tiny"]["eps"]), seed=3)
    write_nvidia_format_pickle(tmp_path / "net.pkl", sd)
    assert "torch_utils" not in sys.modules and "dnnlib" not in sys.modules
    got = load_edm_pickle(tmp_path / "net.pkl")
    assert list(got) == list(sd) or set(got) == set(sd)
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    # a pickle whose reduce would RUN something foreign (here: create a file) loads as inert stand-ins - nothing runs
    import subprocess

    class Evil:
        def __reduce__(self):
            return (subprocess.check_call, (["touch", str(tmp_path / "pwned")],))
    evil = tmp_path / "evil.pkl"
    with open(evil, "wb") as f:
        pickle.dump(dict(ema=Evil()), f)
    assert load_edm_pickle(evil, submodule="") == {}
    assert not (tmp_path / "pwned").exists()


def test_inline_asm_loads_are_never_touched_in_flight(tmp_path):
    """tools/asm_load_audit.py over the compiled gfx950 code of every kernel source that loads into registers from inline asm (the
    sc1 read-backs of the split-K hand-offs): between such a load and a wait that covers it no instruction may read or write its
    destination registers - the register allocator does not know that an asm output is still in flight and has put copies there
    before (conv_halo's CF epilogue, round 5: fixed by making those loads compiler-visible).  hipcc cross-compiles without a GPU."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    root = Path(__file__).resolve().parent.parent
    src = root / "diffusion-nlc_amd" / "csrc"
    stems = ["conv_halo", "conv_fast", "conv_small"]
    procs = [subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", f"-I{root / 'include'}", f"-I{src}", "-S",
                               "--cuda-device-only", "-o", str(tmp_path / f"{s}.s"), str(src / f"{s}.hip")],
                              stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for s in stems]
    for s, pr in zip(stems, procs):
        _, err = pr.communicate(timeout=900)
        assert pr.returncode == 0, f"{s}.hip: {err.decode()[-2000:]}"
    r = subprocess.run([sys.executable, str(root / "tools" / "asm_load_audit.py")] + [str(tmp_path / f"{s}.s") for s in stems],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "0 touches" in r.stdout
