"""Drop-in for the reference's src/unet_simple.py: re-exports the HIP-backed implementation."""
from diffusion_nlc_amd.unet_simple import *  # noqa: F401,F403
from diffusion_nlc_amd import unet_simple as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
