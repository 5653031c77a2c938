"""Reading a Lightning checkpoint of the reference without its Python environment.

``Model.load_from_checkpoint(path, **overrides)`` (sampling.py:54-65, test.py:30-36) reads a file written by
``torch.save({'state_dict': ..., 'hyper_parameters': ...})``.  In the reference's checkpoints the
hyper-parameters contain OmegaConf containers (``spec_args``, ``sampling``, ``training``, ...: Hydra config
nodes passed straight into the constructor), so plain ``torch.load`` needs ``omegaconf`` (and friends)
importable.  Neither is a dependency here: unknown classes are unpickled into inert stand-ins that only
record their state, and OmegaConf containers are then converted to plain dict / list / scalars.

The converter knows the pickled shape of OmegaConf 2.x nodes (containers keep their children in
``_content``, value nodes their payload in ``_val``); it cannot be checked against a real reference
checkpoint in this environment (none is available) and is exercised with synthetic stand-ins in the tests.
"""
from __future__ import annotations

import inspect
import pickle
from typing import Any, Dict

import torch


class _Stub:
    """Inert stand-in for an instance of a class that cannot be imported."""
    _dr_module = "?"
    _dr_name = "?"

    def __init__(self, *args, **kwargs):
        self._dr_args = args
        self._dr_kwargs = kwargs
        self._dr_state = None

    def __setstate__(self, state):
        self._dr_state = state

    def __repr__(self):
        return f"<unpickled {self._dr_module}.{self._dr_name}>"


def _make_stub(module: str, name: str):
    return type(name, (_Stub,), {"_dr_module": module, "_dr_name": name})


class TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return _make_stub(module, name)


class _TolerantPickle:
    """pickle_module for torch.load."""
    __name__ = "diffroll_amd.checkpoint.tolerant_pickle"
    Unpickler = TolerantUnpickler
    UnpicklingError = pickle.UnpicklingError

    @staticmethod
    def load(f, **kwargs):
        return TolerantUnpickler(f, **kwargs).load()


def to_plain(obj: Any) -> Any:
    """Stand-ins / OmegaConf-shaped nodes -> plain Python (dict, list, scalars)."""
    if isinstance(obj, _Stub):
        state = obj._dr_state
        if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):   # (dict, slots) form
            merged = dict(state[0] or {})
            merged.update(state[1])
            state = merged
        if isinstance(state, dict):
            if "_content" in state:
                return to_plain(state["_content"])
            if "_val" in state:
                return to_plain(state["_val"])
            if "_value_" in state:                      # enum members
                return to_plain(state["_value_"])
        if obj._dr_args:                                # e.g. enums reduced to (value,)
            return to_plain(obj._dr_args[0]) if len(obj._dr_args) == 1 else [to_plain(a) for a in obj._dr_args]
        return None
    if isinstance(obj, dict):
        return {to_plain(k) if isinstance(k, _Stub) else k: to_plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_plain(v) for v in obj]
    if hasattr(obj, "items") and not isinstance(obj, (str, bytes)) and not torch.is_tensor(obj):
        try:                                            # a real OmegaConf container when omegaconf IS installed
            return {k: to_plain(v) for k, v in obj.items()}
        except Exception:
            pass
    return obj


def load_checkpoint(path: str) -> Dict[str, Any]:
    """-> {'state_dict': {name: tensor}, 'hyper_parameters': plain dict} (other entries are dropped)."""
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    except Exception:
        ckpt = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_TolerantPickle)
    if not isinstance(ckpt, dict) or "state_dict" not in ckpt:
        raise ValueError(f"{path}: not a Lightning checkpoint (no 'state_dict')")
    return {"state_dict": dict(ckpt["state_dict"]),
            "hyper_parameters": to_plain(ckpt.get("hyper_parameters", {})) or {}}


def constructor_kwargs(cls, hyper_parameters: Dict[str, Any], overrides: Dict[str, Any]) -> Dict[str, Any]:
    """Checkpoint hyper-parameters updated by keyword overrides (which win, as in Lightning), restricted to
    what the constructor accepts."""
    hp = dict(hyper_parameters)
    hp.update(overrides)
    accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
    return {k: v for k, v in hp.items() if k in accepted}
