"""Reading a Lightning checkpoint of the reference without its Python environment.

``Model.load_from_checkpoint(path, **overrides)`` (sampling.py:54-65, test.py:30-36) reads a file written by
``torch.save({'state_dict': ..., 'hyper_parameters': ...})``.  In the reference's checkpoints the
hyper-parameters contain OmegaConf containers (``spec_args``, ``sampling``, ``training``, ...: Hydra config
nodes passed straight into the constructor), so plain ``torch.load`` needs ``omegaconf`` (and friends)
importable.  Neither is a dependency here: unknown classes are unpickled into inert stand-ins that only
record their state, and OmegaConf containers are then converted to plain dict / list / scalars.

The converter knows the pickled shape of OmegaConf 2.x nodes (containers keep their children in
``_content``, value nodes their payload in ``_val``); and resolves
their string interpolations (``sample_rate: ${sampling_rate}``, ``hop_length: ${hop_length}`` in config/spec/mel.yaml
reach the checkpoint unresolved, train_spec_roll.py:30) against the root of the node's ``_parent`` chain.  It cannot
be checked against a real reference checkpoint in this environment (none is available) and is exercised with
synthetic stand-ins in the tests.
"""
from __future__ import annotations

import inspect
import pickle
import re
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


def _state(obj) -> Any:
    """The recorded state of a stand-in as one dict (pickle's (dict, slots) form is merged)."""
    state = obj._dr_state
    if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):
        merged = dict(state[0] or {})
        merged.update(state[1])
        state = merged
    return state


def _root_of(obj) -> Any:
    """OmegaConf nodes keep their container in `_parent`; the root of that chain is where `${a.b}` looks keys up."""
    seen = set()
    while isinstance(obj, _Stub) and id(obj) not in seen:
        seen.add(id(obj))
        st = _state(obj)
        parent = st.get("_parent") if isinstance(st, dict) else None
        if not isinstance(parent, _Stub):
            return obj
        obj = parent
    return obj


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _select(container, dotted: str):
    """Raw (unresolved) node at `a.b.c` below an OmegaConf-shaped container stand-in; KeyError if absent."""
    node = container
    for part in [p for p in dotted.split(".") if p != ""]:
        st = _state(node) if isinstance(node, _Stub) else node
        content = st.get("_content") if isinstance(st, dict) and "_content" in st else st
        if isinstance(content, dict):
            if part not in content:
                raise KeyError(dotted)
            node = content[part]
        elif isinstance(content, (list, tuple)):
            node = content[int(part)]
        else:
            raise KeyError(dotted)
    return node


def _resolve(text: str, node, depth: int = 0) -> Any:
    """OmegaConf interpolation of a value node's string: `${a.b}` is looked up from the ROOT of the node's `_parent`
    chain, `${.a}` / `${..a}` relative to the node's container (one level up per extra dot).  Resolver calls
    (`${now:...}`, `${hydra:...}`) and keys that cannot be found are left as they are - the consumer of that
    value then fails with the unresolved text in its message instead of a silent wrong value."""
    if depth > 16 or "${" not in text:
        return text

    def lookup(key: str):
        if ":" in key:
            raise KeyError(key)
        if key.startswith("."):
            ups = len(key) - len(key.lstrip("."))
            base = node
            for _ in range(ups):
                st = _state(base) if isinstance(base, _Stub) else None
                base = st.get("_parent") if isinstance(st, dict) else None
                if base is None:
                    raise KeyError(key)
            target = _select(base, key.lstrip("."))
        else:
            target = _select(_root_of(node), key)
        return to_plain(target, depth + 1)

    m = _INTERP.fullmatch(text.strip())
    try:
        if m:                                   # the whole value is one interpolation: keep the target's type
            return lookup(m.group(1).strip())
        return _INTERP.sub(lambda mm: str(lookup(mm.group(1).strip())), text)
    except (KeyError, IndexError, ValueError):
        return text


def to_plain(obj: Any, _depth: int = 0) -> Any:
    """Stand-ins / OmegaConf-shaped nodes -> plain Python (dict, list, scalars), interpolations resolved."""
    if isinstance(obj, _Stub):
        state = _state(obj)
        if isinstance(state, dict):
            if "_content" in state:
                return to_plain(state["_content"], _depth)
            if "_val" in state:
                val = state["_val"]
                if isinstance(val, str) and "${" in val:
                    return _resolve(val, obj, _depth)
                return to_plain(val, _depth)
            if "_value_" in state:                      # enum members
                return to_plain(state["_value_"], _depth)
        if obj._dr_args:                                # e.g. enums reduced to (value,)
            return to_plain(obj._dr_args[0], _depth) if len(obj._dr_args) == 1 else [to_plain(a, _depth) for a in obj._dr_args]
        return None
    if isinstance(obj, dict):
        return {to_plain(k, _depth) if isinstance(k, _Stub) else k: to_plain(v, _depth) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_plain(v, _depth) for v in obj]
    if hasattr(obj, "items") and not isinstance(obj, (str, bytes)) and not torch.is_tensor(obj):
        try:                                            # a real OmegaConf container when omegaconf IS installed
            return {k: to_plain(v, _depth) for k, v in obj.items()}
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
