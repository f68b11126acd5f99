"""The FILE-BASED path of the two command-line entry points - what a reference user's first command takes
(/root/reference/image_sample.py:712-800, edm_image_sample.py:110-196): ``store/config/<cfg>.yml``,
``results/<cfg>/<n>/args.json`` of the sigma-net training run, a ``.pt`` eps checkpoint and a sigma checkpoint written with
``torch.save(state_dict)`` under the reference's key names -> ``torch.load`` -> ``load_state_dict`` -> pack -> sample.
Each run must equal the ``--synthetic`` run of the same architecture and weights bit for bit (PNG bytes / sample tensors)."""
import json
import os

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

SIGMA_OVERRIDES = {"final_mlp.weight": 0.1, "final_mlp.bias": 0.5}


def _png_bytes(d):
    return {f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d))}


def _image_sample_tree(root, name, yml, eps_sd, sig_sd, load_folder="3"):
    """<root>/store/config/<name>.yml, <root>/store/models/<name>_eps.pt, <root>/results/<name>/<n>/{args.json, ema_sigma_ckpt_5.pt}"""
    (root / "store" / "config").mkdir(parents=True)
    (root / "store" / "models").mkdir(parents=True)
    run = root / "results" / name / load_folder
    run.mkdir(parents=True)
    with open(root / "store" / "config" / f"{name}.yml", "w") as f:
        yaml.safe_dump(yml, f)
    torch.save(eps_sd, root / "store" / "models" / f"{name}_eps.pt")
    torch.save(sig_sd, run / "ema_sigma_ckpt_5.pt")
    with open(run / "args.json", "w") as f:                    # the fields get_args reads back (image_sample.py:112-121 upstream)
        json.dump(dict(load_eps=f"store/models/{name}_eps.pt", fid_target=None, sigma_block=2, sigma_dropout=0.0,
                       use_sigma_fp16=bool(yml["model"].get("use_fp16", False)), feat_layer=1, lr=1e-4, batch_size=64), f)
    return os.path.join("results", name, load_folder, "ema_sigma_ckpt_5.pt")


def _filled(models, f16_convs=False):
    from diffusion_nlc_amd.filler import fill_state_dict
    eps, sig = models
    sd_e = fill_state_dict(eps.state_dict(), seed=0)
    sd_s = fill_state_dict(sig.state_dict(), seed=1, overrides=SIGMA_OVERRIDES)
    if f16_convs:
        # a checkpoint saved from a module after convert_to_fp16() (src/fp16_util.py:15-22 halves the conv weights of the three
        # block lists): half tensors in the file, cast back by load_state_dict - the f16 rounding is the one the f16 pack applies
        # (the qkv projections stay f32 in this file: their rows are packed with the attention scale folded in, in f64 before the one
        #  rounding to f16 - a pre-rounded weight would be rounded twice and differ in the last bit from the --synthetic run)
        for k, v in sd_e.items():
            if k.split(".")[0] in ("input_blocks", "middle_block", "output_blocks") and k.endswith(".weight") and v.dim() >= 3 \
                    and ".qkv." not in k:
                sd_e[k] = v.half()
    return sd_e, sd_s


@pytest.mark.parametrize("synthetic,preset,f16", [("cifar_tiny", "celeba_hq", False), ("adm_tiny", "imagenet", True)], ids=["simple", "adm-fp16"])
def test_image_sample_from_files_equals_the_synthetic_run(tmp_path, monkeypatch, synthetic, preset, f16):
    import image_sample
    from src.script_util import create_sigma_eps_model, create_simple_sigma_eps_model
    monkeypatch.chdir(tmp_path)
    common = ["--batch_size", "2", "--sample_size", "4", "--max_T", "5", "--seed", "7", "--method", "pred_denoise_base", "--eta", "0"]
    # (a) the built-in configuration with filler weights
    a_args, a_cfg = image_sample.get_args(["--synthetic", synthetic, "--save_folder", str(tmp_path / "syn"), *common])
    image_sample.main(a_args, a_cfg)
    # (b) the same architecture and weights through files
    yml = image_sample.SYNTHETIC[synthetic][0]
    mc = a_cfg.model
    models = (create_sigma_eps_model(**vars(mc)) if mc.type == "openai" else create_simple_sigma_eps_model(a_cfg))[:2]
    name = f"{synthetic}_files"
    sigma_ckpt = _image_sample_tree(tmp_path, name, yml, *_filled(models, f16_convs=f16))
    b_args, b_cfg = image_sample.get_args(["--config", preset, "--config_path", name, "--load_folder", "3", "--load_sigma", sigma_ckpt,
                                           "--save_folder", str(tmp_path / "files"), *common])
    assert b_args.synthetic is None and b_args.load_eps == f"store/models/{name}_eps.pt"
    assert b_args.result_dir == os.path.join("results", name, "3")
    assert vars(b_cfg.model) == vars(a_cfg.model) and b_args.norm_max == a_args.norm_max and b_args.clip_fn == a_args.clip_fn
    out = image_sample.main(b_args, b_cfg)
    assert "fid" in out
    with open(tmp_path / "files" / "args.json") as f:
        saved = json.load(f)
    assert saved["load_eps"] == b_args.load_eps and saved["config_path"] == name
    assert os.path.exists(tmp_path / "files" / "0" / "results.json")
    a_png, b_png = _png_bytes(tmp_path / "syn" / "0" / "images"), _png_bytes(tmp_path / "files" / "0" / "images")
    assert list(a_png) == ["00-00000-000.png", "00-00000-001.png", "00-00001-000.png", "00-00001-001.png"]
    assert a_png == b_png                                               # same weights, same seeds: the same bytes


def test_image_sample_missing_files_fail_like_upstream(tmp_path, monkeypatch):
    import image_sample
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError):
        image_sample.get_args(["--config", "cifar10"])                 # results/cifar10_adm/7/args.json does not exist here


def test_edm_image_sample_from_files_equals_the_synthetic_run(tmp_path, monkeypatch):
    import edm_image_sample
    from diffusion_nlc_amd.filler import fill_state_dict
    from src.script_util import create_edm_sigma_eps_model
    monkeypatch.chdir(tmp_path)
    common = ["--sampler", "edm", "--batch_size", "2", "--sample_size", "4", "--num_timesteps", "4", "--device", "cuda:0", "--save_png", "0"]
    a_args, a_cfg = edm_image_sample.get_args(["--synthetic", "tiny", "--save_folder", str(tmp_path / "syn"), *common])
    _, a_samples = edm_image_sample.main(a_args, a_cfg, return_samples=True)
    eps, sig, _ = create_edm_sigma_eps_model(**vars(a_cfg.model))
    tmpl = eps.state_dict()
    for k in tmpl:
        if k.endswith("resample_filter"):
            tmpl[k] = torch.ones_like(tmpl[k]) / 4.0
    (tmp_path / "store" / "config").mkdir(parents=True)
    (tmp_path / "store" / "models").mkdir(parents=True)
    run = tmp_path / "results" / "cifar10" / "6"
    run.mkdir(parents=True)
    with open(tmp_path / "store" / "config" / "cifar10.yml", "w") as f:
        yaml.safe_dump(edm_image_sample.SYNTHETIC["tiny"], f)
    torch.save(fill_state_dict(tmpl, seed=0), tmp_path / "store" / "models" / "edm_tiny.pt")
    torch.save(fill_state_dict(sig.state_dict(), seed=1, overrides=SIGMA_OVERRIDES), run / "ema_sigma_ckpt_100.pt")
    with open(run / "args.json", "w") as f:
        json.dump(dict(load_eps="store/models/edm_tiny.pt", fid_target=None, sigma_block=2, sigma_dropout=0.0, use_sigma_fp16=False,
                       feat_layer=1), f)
    b_args, b_cfg = edm_image_sample.get_args(["--config", "cifar10", "--load_sigma", "results/cifar10/6/ema_sigma_ckpt_100.pt",
                                               "--save_folder", str(tmp_path / "files"), *common])
    assert b_args.synthetic is None and b_args.load_eps == "store/models/edm_tiny.pt" and b_args.norm_max == 54.63
    assert vars(b_cfg.model) == vars(a_cfg.model)
    log, b_samples = edm_image_sample.main(b_args, b_cfg, return_samples=True)
    assert set(log) == {"fid"} and os.path.exists(tmp_path / "files" / "0" / "results.json")
    assert b_samples.shape == (4, 3, 32, 32) and torch.equal(a_samples.cpu(), b_samples.cpu())
    # ... and through an NVIDIA-format network pickle (edm_image_sample.py:152-156 upstream: pickle.load(f)['ema'], .model.state_dict())
    from tests.test_host_cpu import write_nvidia_format_pickle
    write_nvidia_format_pickle(tmp_path / "store" / "models" / "edm_tiny.pkl", fill_state_dict(tmpl, seed=0))
    with open(run / "args.json", "w") as f:
        json.dump(dict(load_eps="store/models/edm_tiny.pkl", fid_target=None, sigma_block=2, sigma_dropout=0.0, use_sigma_fp16=False,
                       feat_layer=1), f)
    c_args, c_cfg = edm_image_sample.get_args(["--config", "cifar10", "--load_sigma", "results/cifar10/6/ema_sigma_ckpt_100.pt",
                                               "--save_folder", str(tmp_path / "pkl"), *common])
    _, c_samples = edm_image_sample.main(c_args, c_cfg, return_samples=True)
    assert torch.equal(a_samples.cpu(), c_samples.cpu())
