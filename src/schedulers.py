"""Drop-in for the reference's src/schedulers.py: re-exports the HIP-backed implementation."""
from diffusion_nlc_amd.schedulers import *  # noqa: F401,F403
from diffusion_nlc_amd import schedulers as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
