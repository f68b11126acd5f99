"""Drop-in for the reference's src/unet_adm.py: re-exports the HIP-backed implementation."""
from diffusion_nlc_amd.unet_adm import *  # noqa: F401,F403
from diffusion_nlc_amd import unet_adm as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
