"""diffusion-nlc_amd: MI355X-native DDIM/EDM + noise-level-correction sampling hot path.

Import it as ``diffusion_nlc_amd`` (the importable alias next to this directory): the
directory name carries a hyphen, so ``diffusion_nlc_amd/__init__.py`` points its ``__path__``
here.  The package holds the HIP kernels + C ABI (``csrc/``, ``libnlc_hip.so``), the ctypes
binding (``_ext``), tensor shims (``ops``) and the host-side mirror of the reference's
network / scheduler / sampling-loop interface.
"""
__version__ = "0.1.0"
