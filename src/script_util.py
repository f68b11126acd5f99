"""Drop-in for the reference's src/script_util.py: re-exports the HIP-backed implementation."""
from diffusion_nlc_amd.script_util import *  # noqa: F401,F403
from diffusion_nlc_amd import script_util as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
