"""Importable alias of the ``diffusion-nlc_amd/`` package directory (hyphens cannot be imported)."""
from pathlib import Path as _Path

_real = _Path(__file__).resolve().parent.parent / "diffusion-nlc_amd"
__path__ = [str(_real)]
exec(compile((_real / "__init__.py").read_text(), str(_real / "__init__.py"), "exec"))
