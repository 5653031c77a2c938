"""Reading a Lightning checkpoint of the reference without its Python environment.

``Model.load_from_checkpoint(path, **overrides)`` (sampling.py:54-65, test.py:30-36) reads a file written by
``torch.save({'state_dict': ..., 'hyper_parameters': ...})``.  In the reference's checkpoints the
hyper-parameters contain OmegaConf containers (``spec_args``, ``sampling``, ``training``, ...: Hydra config
nodes passed straight into the constructor), so plain ``torch.load`` needs ``omegaconf`` (and friends)
importable.  Neither is a dependency here, and NOTHING the file names is trusted: apart from an allow-list of tensor /
container / scalar constructors every global of the pickle stream is unpickled into an inert stand-in that only records
its state (never imported, never called), and OmegaConf containers are then converted to plain dict / list / scalars.

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
    # (class-level defaults: pickle's NEWOBJ path - every plain object pickled with protocol >= 2 - creates the instance
    # with __new__ and never runs __init__)
    _dr_args = ()
    _dr_kwargs = {}
    _dr_state = None

    def __new__(cls, *args, **kwargs):
        self = object.__new__(cls)
        self._dr_args = args
        self._dr_kwargs = kwargs
        return self

    def __init__(self, *args, **kwargs):
        self._dr_args = args
        self._dr_kwargs = kwargs

    def __setstate__(self, state):
        self._dr_state = state

    def __repr__(self):
        return f"<unpickled {self._dr_module}.{self._dr_name}>"


def _make_stub(module: str, name: str):
    return type(name, (_Stub,), {"_dr_module": module, "_dr_name": name})


# What a checkpoint's pickle stream may IMPORT: the constructors of tensors, containers and scalars - nothing else.  Every
# other global the stream names (pytorch_lightning / omegaconf classes, but equally os.system, builtins.eval, ...) becomes an
# inert stand-in that records its arguments: it is never imported, never called.
_SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray", "complex",
                  "slice", "range"}
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"),
    ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
    ("torch", "Size"), ("torch", "device"),
    ("numpy", "dtype"), ("numpy", "ndarray"),
    ("numpy.core.multiarray", "scalar"), ("numpy.core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "scalar"), ("numpy._core.multiarray", "_reconstruct"),
    ("_codecs", "encode"),
}


def _allowed(module: str, name: str) -> bool:
    if (module, name) in _SAFE_GLOBALS:
        return True
    if module == "builtins" and name in _SAFE_BUILTINS:
        return True
    if module == "torch" and name.isidentifier() and isinstance(getattr(torch, name, None), torch.dtype):
        return True                                     # torch.float32, ... (a dtype pickles as a global of that name)
    return False


class TolerantUnpickler(pickle.Unpickler):
    """Allow-listed globals are imported; everything else unpickles into an inert stand-in (nothing of the checkpoint's
    packages - or of anybody's - is imported or executed)."""

    def find_class(self, module, name):
        if _allowed(module, name):
            return super().find_class(module, name)
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
            if _depth <= 16:                            # any other object (argparse.Namespace, a dataclass): its attributes
                return {k: to_plain(v, _depth + 1) for k, v in state.items() if isinstance(k, str) and not k.startswith("_")}
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


def load_checkpoint(path: str, trust: bool = False) -> Dict[str, Any]:
    """-> {'state_dict': {name: tensor}, 'hyper_parameters': plain dict} (other entries are dropped).

    Safe by default: first ``torch.load(weights_only=True)``; a file that needs more than that (every Lightning checkpoint
    with OmegaConf hyper-parameters does) is read by an allow-listing unpickler - tensor / container / scalar
    constructors are imported, EVERY other global the pickle names becomes an inert stand-in: a checkpoint that names
    ``os.system`` loads to a stub and executes nothing.  ``trust=True`` opts into the full unpickle (what Lightning's own
    ``load_from_checkpoint``, sampling.py:54, does): only for files from sources you would run code from."""
    if trust:
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    else:
        try:
            ckpt = torch.load(path, map_location="cpu", weights_only=True)
        except Exception:       # noqa: BLE001 - whatever the strict loader rejects goes through the allow-list
            ckpt = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_TolerantPickle)
    if not isinstance(ckpt, dict) or "state_dict" not in ckpt:
        raise ValueError(f"{path}: not a Lightning checkpoint (no 'state_dict')")
    sd = ckpt["state_dict"]
    bad = [k for k, v in sd.items() if not torch.is_tensor(v)] if isinstance(sd, dict) else ["state_dict"]
    if bad:
        raise ValueError(f"{path}: state_dict entries that are not tensors ({bad[:3]}): refused "
                         "(pass trust=True to unpickle the file as Lightning would)")
    return {"state_dict": dict(sd), "hyper_parameters": to_plain(ckpt.get("hyper_parameters", {})) or {}}


def constructor_kwargs(cls, hyper_parameters: Dict[str, Any], overrides: Dict[str, Any]) -> Dict[str, Any]:
    """Checkpoint hyper-parameters updated by keyword overrides (which win, as in Lightning), restricted to
    what the constructor accepts."""
    hp = dict(hyper_parameters)
    hp.update(overrides)
    accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
    return {k: v for k, v in hp.items() if k in accepted}
