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
    import edm_image_sample
    args = edm_image_sample.get_args(["--synthetic", "tiny", "--batch_size", "2", "--sample_size", "4", "--num_timesteps", "4",
                                      "--save_folder", str(tmp_path)])
    log, samples = edm_image_sample.main(args)
    assert samples.shape == (4, 3, 32, 32) and torch.isfinite(samples).all()
    assert "fid" in log
