"""The reference's command-line entry points (image_sample.py / edm_image_sample.py) with their own flags,
run in-process on the GPU with the built-in synthetic configurations."""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(tmp_path, *flags):
    import image_sample
    argv = ["--synthetic", "cifar_tiny", "--batch_size", "2", "--sample_size", "4", "--save_png", "0", "--save_folder", str(tmp_path),
            "--max_T", "6", "--seed", "7", *flags]
    args, config = image_sample.get_args(argv)
    out = image_sample.main(args, config)
    with open(os.path.join(str(tmp_path), "0", "results.json")) as f:
        saved = json.load(f)
    return args, out, saved


def test_image_sample_denoise_preset(tmp_path):
    args, out, saved = _run(tmp_path, "--method", "pred_denoise_base", "--eta", "0")
    assert args.sampling == "denoise" and args.num_timesteps == 6 and not args.continuous_t
    assert "fid" in out and "fid" in saved


def test_image_sample_projection_preset(tmp_path):
    """--method pred_proj: Linear sigma spacing, continuous t, redesigned sigma tail, projection_loop."""
    args, out, _ = _run(tmp_path, "--method", "pred_proj", "--num_timesteps", "4", "--eta", "0", "--sigma_estimate", "0100",
                        "--end_sigma", "0.01", "--cycle_size", "1")
    assert args.sampling == "project" and args.continuous_t and args.redesign_sigma and args.sigma_estimate_rate == [0.0, 1.0, 0.0, 0.0]
    assert "fid" in out


def test_image_sample_inpainting(tmp_path):
    args, out, saved = _run(tmp_path, "--method", "pred_denoise_base", "--eta", "0", "--constraint", "inpainting",
                            "--clip_fn", "clamp")
    assert saved["const_f_loss"] < 1e-3                       # known pixels are reproduced
    assert math.isfinite(saved["psner"]) and saved["psner"] > 3.0   # half the pixels are exact: PSNR(U(0,1) vs noise) + 3 dB
    assert len(saved["full_log"]["psnr"]) >= 4


def test_edm_image_sample_entry(tmp_path):
    """The reference's own invocation line (`--sampler euler --start_sigma 80 --end_sigma 0.002`, edm_image_sample.py:23,28-29)
    plus the built-in configuration: main() returns the reference's log_dict, writes args.json / results.json / the PNGs
    (edm_image_sample.py:115-137,181-196) and a second run skips the batches whose PNGs exist (src/experiments.py:935-943)."""
    import edm_image_sample
    argv = ["--sampler", "euler", "--start_sigma", "80", "--end_sigma", "0.002", "--synthetic", "tiny", "--batch_size", "2",
            "--sample_size", "4", "--num_timesteps", "4", "--save_folder", str(tmp_path / "run"), "--device", "cuda:0"]
    args, config = edm_image_sample.get_args(argv)
    assert args.norm_max == 54.63 and args.norm_min == 0 and args.test_dir == os.path.join("temp", "cifar10")
    log = edm_image_sample.main(args, config)
    assert isinstance(log, dict) and set(log) == {"fid"}
    run = tmp_path / "run"
    with open(run / "args.json") as f:
        saved_args = json.load(f)
    assert saved_args["sampler"] == "euler" and saved_args["start_sigma"] == 80.0 and saved_args["device"] == "cuda:0"
    with open(run / "0" / "results.json") as f:
        assert set(json.load(f)) == {"fid"}
    pngs = sorted(os.listdir(run / "0" / "images"))
    assert pngs == ["00-00000-000.png", "00-00000-001.png", "00-00001-000.png", "00-00001-001.png"]
    from PIL import Image
    assert Image.open(run / "0" / "images" / pngs[0]).size == (32, 32)
    # second run into the same folder: every batch is skipped, nothing is sampled
    args2, config2 = edm_image_sample.get_args(argv)
    log2, samples2 = edm_image_sample.main(args2, config2, return_samples=True)
    assert samples2.shape[0] == 0 and "fid" in log2
    # --sample_overwrite 1 resamples; Heun (--sampler edm) differs from Euler
    args3, config3 = edm_image_sample.get_args(argv + ["--sample_overwrite", "1", "--sampler", "edm", "--save_png", "0"])
    log3, samples3 = edm_image_sample.main(args3, config3, return_samples=True)
    assert samples3.shape == (4, 3, 32, 32) and torch.isfinite(samples3).all()
    assert float(samples3.min()) >= 0.0 and float(samples3.max()) <= 1.0


def test_evaluate_edm_signature_and_return(tmp_path):
    """EDMImageExperiment.evaluate_edm(n_samples, images_dir, ...) as the reference's CLI calls it (edm_image_sample.py:189-195):
    positional images_dir, returns log_dict; PNG pixels are the rounded samples."""
    import inspect
    import numpy as np
    from PIL import Image
    from src.experiments import EDMImageExperiment
    from tests.test_host_cpu import build_product
    from tests.util import state_dicts
    sig = inspect.signature(EDMImageExperiment.evaluate_edm)
    names = list(sig.parameters)
    assert names[:13] == ["self", "n_samples", "images_dir", "gen", "style", "norm_eps", "refine_prior_sigma", "microbatch",
                          "sigma_scheduler", "eps_ratio", "eps_scale", "use_second_order", "return_samples"]
    assert sig.parameters["images_dir"].default is inspect.Parameter.empty
    eps, sgm, _ = build_product("edm_tiny")
    e, s = state_dicts("edm_tiny")
    eps.load_state_dict(e); sgm.load_state_dict(s)
    eps.to("cuda:0"); sgm.to("cuda:0")
    exp = EDMImageExperiment(eps, None, batch_size=2, data_shape=(3, 32, 32), device="cuda:0", num_timesteps=3)
    exp.set_model(eps, sgm, learn_epsvar=False)
    exp.fid_helper(None)
    exp.set_norm_maxmin(0, 54.63)
    d = tmp_path / "img"
    d.mkdir()
    log = exp.evaluate_edm(2, str(d), gen=exp.new_gen(), style="pred_partial,pred", norm_eps="00")
    assert isinstance(log, dict) and "fid" in log
    with open(tmp_path / "r.json", "w") as f:
        json.dump(log, f)                                           # what the reference's main does with it (:194-195)
    img = np.asarray(Image.open(d / "00-00000-001.png")).astype(np.int64)
    want = (exp.last_samples[1].float().cpu() * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy().astype(np.int64)
    assert np.abs(img - want).max() <= 1
    with pytest.raises(ValueError):
        exp.evaluate_edm(3, str(d))
