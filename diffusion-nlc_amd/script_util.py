"""Model factories with the reference's signatures (src/script_util.py:136-270).

``create_sigma_eps_model`` / ``create_simple_sigma_eps_model`` / ``create_edm_sigma_eps_model``
return ``(eps_model, sigma_model, feat_shape)`` exactly as the reference does, built from the
HIP-backed networks of this package.
"""
from __future__ import annotations

NUM_CLASSES = 1000          # src/script_util.py:12


def create_sigma_eps_model(image_size, num_channels, num_res_blocks, channel_mult="", learn_sigma=False,
                           class_cond=False, use_checkpoint=False, attention_resolutions="16", num_heads=1,
                           num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0.0,
                           resblock_updown=False, use_fp16=False, use_new_attention_order=False, sigma_block=2,
                           sigma_dropout=0.0, use_sigma_fp16=False, **kwargs):
    """src/script_util.py:136-206.  NB: like the reference, ``feat_layer`` in kwargs is ignored (ADM always
    encodes through the full middle block)."""
    from .unet_adm import SigmaModel, UNetModel
    if channel_mult == "":
        table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4), 32: (1, 2, 2, 2)}
        if image_size not in table:
            raise ValueError(f"unsupported image size: {image_size}")
        channel_mult = table[image_size]
    else:
        channel_mult = tuple(int(c) for c in channel_mult.split(","))
    attention_ds = tuple(image_size // int(res) for res in attention_resolutions.split(","))
    eps_model = UNetModel(image_size=image_size, in_channels=3, model_channels=num_channels,
                          out_channels=(3 if not learn_sigma else 6), num_res_blocks=num_res_blocks,
                          attention_resolutions=attention_ds, dropout=dropout, channel_mult=channel_mult,
                          num_classes=(NUM_CLASSES if class_cond else None), use_checkpoint=use_checkpoint, use_fp16=use_fp16, num_heads=num_heads,
                          num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
                          use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
                          use_new_attention_order=use_new_attention_order)
    inp_channels = int(num_channels * channel_mult[-1])
    inp_dim = int(image_size * 0.5 ** (len(channel_mult) - 1))
    sigma_model = SigmaModel(dim=inp_dim, channels=inp_channels, n_blocks=sigma_block, out_dim=1, dropout=sigma_dropout,
                             num_heads=num_heads, num_head_channels=num_head_channels,
                             use_new_attention_order=use_new_attention_order, use_checkpoint=use_checkpoint,
                             use_fp16=use_sigma_fp16)
    return eps_model, sigma_model, (inp_channels, inp_dim, inp_dim)


def create_simple_sigma_eps_model(config):
    """src/script_util.py:209-219 (``config`` is the nested Namespace the entry point builds from YAML)."""
    from .unet_simple import Model, SigmaModel
    eps_model = Model(config)
    num_channels, channel_mult = config.model.ch, tuple(config.model.ch_mult)
    inp_channels = int(num_channels * channel_mult[-1])
    inp_dim = int(config.data.image_size * 0.5 ** (len(channel_mult) - 1))
    sigma_model = SigmaModel(dim=inp_dim, channels=inp_channels, n_blocks=config.model.sigma_block, out_dim=1,
                             dropout=config.model.sigma_dropout)
    return eps_model, sigma_model, (inp_channels, inp_dim, inp_dim)


def create_edm_sigma_eps_model(img_resolution, in_channels, out_channels, augment_dim=0, model_channels=128,
                               channel_mult=[1, 2, 2, 2], channel_mult_emb=4, num_blocks=4, attn_resolutions=[16],
                               dropout=0.10, embedding_type="positional", encoder_type="standard",
                               decoder_type="standard", resample_filter=[1, 1], sigma_block=2, sigma_dropout=0.0, **kwargs):
    """src/script_util.py:222-270."""
    from .edm_networks import SigmaModel, SongUNet
    eps_model = SongUNet(img_resolution=img_resolution, in_channels=in_channels, out_channels=out_channels, label_dim=0,
                         augment_dim=augment_dim, model_channels=model_channels, channel_mult=channel_mult,
                         channel_mult_emb=channel_mult_emb, num_blocks=num_blocks, attn_resolutions=attn_resolutions,
                         dropout=dropout, embedding_type=embedding_type, channel_mult_noise=1, encoder_type=encoder_type,
                         decoder_type=decoder_type, resample_filter=resample_filter)
    inp_channels = int(model_channels * channel_mult[-1])
    inp_dim = int(img_resolution * 0.5 ** (len(channel_mult) - 1))
    sigma_model = SigmaModel(dim=inp_dim, channels=inp_channels, n_blocks=sigma_block, out_dim=1, dropout=sigma_dropout,
                             resample_filter=resample_filter)
    return eps_model, sigma_model, (inp_channels, inp_dim, inp_dim)
