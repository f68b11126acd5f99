"""Import-path shim: ``src.constraint_functions`` of the reference -> the HIP implementation."""
from diffusion_nlc_amd.constraint_functions import *  # noqa: F401,F403
from diffusion_nlc_amd import constraint_functions as _impl

svd_constraint = _impl.svd_constraint
Constraint_Function = _impl.Constraint_Function
Inpainting = _impl.Inpainting
