"""Drop-in for the reference's src/experiments.py: re-exports the HIP-backed implementation."""
from diffusion_nlc_amd.experiments import *  # noqa: F401,F403
from diffusion_nlc_amd import experiments as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
