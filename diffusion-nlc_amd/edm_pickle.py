"""Read the weights out of an NVIDIA EDM network pickle (`edm-cifar10-32x32-uncond-vp.pkl`, ...) without the `dnnlib` / `torch_utils`
packages the reference needs for it (/root/reference/edm_image_sample.py:152-156: ``pickle.load(f)['ema']`` then
``saved_eps.model.state_dict()``).

Those pickles store every `persistence.persistent_class` instance as a call of ``torch_utils.persistence._reconstruct_persistent_obj``
on one dict - ``{type: 'class', version, module_src, class_name, state}`` where ``state`` is the instance's ``__dict__`` (for an
``nn.Module``: ``_parameters``, ``_buffers``, ``_modules``, ...) - and the stock hook re-creates the class by EXECUTING ``module_src``.
Nothing here executes it: the hook is replaced by a stand-in that keeps ``state``, and the state_dict is collected by walking
``_parameters`` / ``_buffers`` / ``_modules`` exactly as ``nn.Module.state_dict`` does.  Only torch / numpy / collections / builtins
globals are imported; every other global the pickle names becomes an inert stand-in that records its state or arguments (the stock
``pickle.load`` upstream imports and runs anything).
"""
from __future__ import annotations

import io
import pickle
from collections import OrderedDict

import torch


class PersistentStub:
    """What a persistent_class instance leaves behind here: its class name and its pickled ``__dict__`` (attribute access works)."""

    def __init__(self, meta):
        self.__dict__.update(meta.get("state") or {})
        self.__dict__["_nlc_class_name"] = meta.get("class_name")

    def __getattr__(self, name):                     # nn.Module's lookup order for what is not a plain attribute
        for table in ("_parameters", "_buffers", "_modules"):
            t = self.__dict__.get(table)
            if t is not None and name in t:
                return t[name]
        raise AttributeError(name)


class InertStub(PersistentStub):
    """Stand-in for a global this reader does not import: ``cls.__new__`` + state (pickle's NEWOBJ / BUILD) or ``cls(*args)`` (REDUCE)
    just keep what they are given."""

    def __init__(self, *args, **kwargs):
        self.__dict__["_nlc_args"] = (args, kwargs)

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):      # (dict state, slots state)
            state = {**(state[0] or {}), **state[1]}
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_nlc_state"] = state


class _EasyDict(dict):
    """dnnlib.util.EasyDict: a dict with attribute access."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _reconstruct(meta):
    if meta.get("type") != "class":
        raise pickle.UnpicklingError(f"EDM pickle: unknown persistent object type {meta.get('type')!r}")
    return PersistentStub(meta)


_ALLOWED_ROOTS = ("torch", "numpy", "collections", "builtins", "_codecs", "copyreg")


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "torch_utils.persistence" and name == "_reconstruct_persistent_obj":
            return _reconstruct
        if module.split(".")[0] == "dnnlib" and name == "EasyDict":
            return _EasyDict
        if module.split(".")[0] in _ALLOWED_ROOTS:
            return super().find_class(module, name)
        # anything else - a network class that is NOT a persistent_class (this repository's reference pickles its SongUNet by
        # reference, src/edm_networks.py:732), a helper function - becomes an inert stand-in: instantiating or "calling" it only
        # records its state / arguments, nothing foreign is imported or run
        return type(name, (InertStub,), {"_nlc_global": f"{module}.{name}"})


def _walk(obj, prefix, out):
    d = obj.__dict__
    for k, v in (d.get("_parameters") or {}).items():
        if v is not None:
            out[prefix + k] = v.detach() if torch.is_tensor(v) else torch.as_tensor(v)
    skip = d.get("_non_persistent_buffers_set") or set()
    for k, v in (d.get("_buffers") or {}).items():
        if v is not None and k not in skip:
            out[prefix + k] = v.detach() if torch.is_tensor(v) else torch.as_tensor(v)
    for k, m in (d.get("_modules") or {}).items():
        if m is not None:
            _walk(m, prefix + k + ".", out)


def state_dict_of(obj) -> "OrderedDict[str, torch.Tensor]":
    """``obj.state_dict()`` for a PersistentStub / nn.Module tree as the pickle left it."""
    out = OrderedDict()
    _walk(obj, "", out)
    return out


def load_edm_pickle(path_or_file, which: str = "ema", submodule: str = "model") -> "OrderedDict[str, torch.Tensor]":
    """The state_dict the reference loads from an EDM pickle: ``pickle.load(f)[which].<submodule>.state_dict()`` (EDMPrecond.model = the
    SongUNet), tensors moved to the CPU."""
    if hasattr(path_or_file, "read"):
        data = path_or_file.read()
    else:
        with open(path_or_file, "rb") as f:
            data = f.read()
    top = _Unpickler(io.BytesIO(data)).load()
    net = top[which] if isinstance(top, dict) else top
    if submodule:
        net = getattr(net, submodule) if not isinstance(net, dict) else net[submodule]
    return OrderedDict((k, v.cpu()) for k, v in state_dict_of(net).items())
