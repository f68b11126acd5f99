"""Drop-in for the reference's src/edm_networks.py: re-exports the HIP-backed implementation."""
from diffusion_nlc_amd.edm_networks import *  # noqa: F401,F403
from diffusion_nlc_amd import edm_networks as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
