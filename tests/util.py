"""Shared test helpers: fixtures, synthetic state_dicts and oracle closures for the tiny nets."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"
_DT = {"float32": torch.float32, "int64": torch.int64, "float64": torch.float64}


def load_specs():
    return json.loads((GOLDEN / "specs.json").read_text())


def load_npz(name):
    z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        if k in ("cfg", "_meta"):
            out[k] = json.loads(str(v))
        else:
            out[k] = torch.from_numpy(np.array(v))
    return out


def template_from_spec(spec):
    """{key: [shape, dtype]} (recorded from the reference module) -> zero tensors of those shapes."""
    t = {}
    for k, (shape, dt) in spec.items():
        if k.endswith("resample_filter"):
            t[k] = torch.ones(shape, dtype=_DT[dt]) / 4.0       # f.ger(f)/f.sum()^2 with f=[1,1] (edm_networks.py:69-71)
        else:
            t[k] = torch.zeros(shape, dtype=_DT[dt])
    return t


def state_dicts(tag):
    from diffusion_nlc_amd.filler import fill_state_dict
    specs = load_specs()
    eps = fill_state_dict(template_from_spec(specs[tag]["eps"]), seed=0)
    sig = fill_state_dict(template_from_spec(specs[tag]["sigma"]), seed=1, overrides=specs["_configs"]["sigma_overrides"])
    return eps, sig


def oracle_nets(tag):
    """(eps_fn, encode_fn, sigma_fn, both_fn) closures over the oracle restatement for a tiny config."""
    from oracle import adm, edm, simple
    cfgs = load_specs()["_configs"]
    sd_e, sd_s = state_dicts(tag)
    if tag.startswith("adm"):
        ucfg, scfg, _ = adm.configs_from_factory(**cfgs[tag])
        return (lambda x, t: adm.unet(sd_e, ucfg, x, t, "forward"), lambda x, t: adm.unet(sd_e, ucfg, x, t, "encode"),
                lambda f: adm.sigma_net(sd_s, scfg, f), lambda x, t: adm.unet(sd_e, ucfg, x, t, "both"))
    if tag.startswith("simple"):
        c = cfgs[tag]
        cfg = simple.SimpleConfig(ch=c["ch"], out_ch=c["out_ch"], ch_mult=tuple(c["ch_mult"]), num_res_blocks=c["num_res_blocks"],
                                  attn_resolutions=tuple(c["attn_resolutions"]), in_channels=c["in_channels"],
                                  resolution=c["image_size"], resamp_with_conv=c["resamp_with_conv"], feat_layer=c["feat_layer"],
                                  sigma_block=c["sigma_block"])
        _, dim = simple.sigma_dims(cfg)
        return (lambda x, t: simple.unet(sd_e, cfg, x, t, "forward"), lambda x, t: simple.unet(sd_e, cfg, x, t, "encode"),
                lambda f: simple.sigma_net(sd_s, dim, cfg.sigma_block, f), lambda x, t: simple.unet(sd_e, cfg, x, t, "both"))
    if tag.startswith("edm"):
        c = cfgs[tag]
        cfg = edm.EdmConfig(img_resolution=c["img_resolution"], in_channels=c["in_channels"], out_channels=c["out_channels"],
                            augment_dim=c["augment_dim"], model_channels=c["model_channels"], channel_mult=tuple(c["channel_mult"]),
                            num_blocks=c["num_blocks"], attn_resolutions=tuple(c["attn_resolutions"]), sigma_block=c["sigma_block"])
        _, dim = edm.sigma_dims(cfg)
        return (lambda x, t: edm.unet(sd_e, cfg, x, t, "forward"), lambda x, t: edm.unet(sd_e, cfg, x, t, "encode"),
                lambda f: edm.sigma_net(sd_s, dim, cfg.sigma_block, f), None)
    raise KeyError(tag)


def max_err(a, b):
    return (a.double() - b.double()).abs().max().item()
