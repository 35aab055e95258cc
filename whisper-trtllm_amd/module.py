"""Parameter tree containers with the reference's attribute surface.

The reference's build scripts assign checkpoints through attribute paths such as
`model.layers[i].self_attn.qkv.weight.value = ndarray` (build_encoder.py:71-91) and hand
`model.named_parameters()` to the network (build_encoder.py:100).  These classes keep exactly that
surface (tensorrt_llm/module.py:8-164, parameter.py:11-59) — there is no graph tracing behind them: a
"forward" only registers the module with the active Network so that `Builder.build_engine` can pack it.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Iterator, Optional, Sequence, Tuple

import numpy as np


class Parameter:
    """A named fp32 ndarray slot; assigning `.value` checks the shape like parameter.py:50-55."""

    def __init__(self, value: Optional[np.ndarray] = None, shape: Optional[Sequence[int]] = None, dtype="float32"):
        if value is None:
            if shape is None:
                raise ValueError("Parameter needs a value or a shape")
            value = np.zeros(tuple(shape), dtype=np.float32)
        self._value = np.asarray(value)
        self.dtype = dtype

    @property
    def value(self) -> np.ndarray:
        return self._value

    @value.setter
    def value(self, v):
        v = np.asarray(v)
        assert v.shape == self._value.shape, \
            f"The value updated is not the same shape as the original. Updated: {v.shape}, original: {self._value.shape}"
        self._value = v

    @property
    def shape(self):
        return self._value.shape


class Module:
    def __init__(self):
        object.__setattr__(self, "_modules", OrderedDict())
        object.__setattr__(self, "_parameters", OrderedDict())

    def __setattr__(self, name, value):
        if isinstance(value, Parameter):
            self._parameters[name] = value
        elif isinstance(value, Module):
            self._modules[name] = value
        object.__setattr__(self, name, value)

    def register_parameter(self, name: str, param: Optional[Parameter]):
        if param is not None:
            self._parameters[name] = param
        object.__setattr__(self, name, param)

    def named_children(self):
        return iter(self._modules.items())

    def named_modules(self, prefix: str = "") -> Iterator[Tuple[str, "Module"]]:
        yield prefix, self
        for name, m in self._modules.items():
            yield from m.named_modules(prefix + ("." if prefix else "") + name)

    def named_parameters(self, prefix: str = "") -> Iterator[Tuple[str, Parameter]]:
        for mod_name, m in self.named_modules(prefix):
            for pname, p in m._parameters.items():
                if p is not None:
                    yield (mod_name + "." if mod_name else "") + pname, p

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)


class ModuleList(Module):
    def __init__(self, modules):
        super().__init__()
        for i, m in enumerate(modules):
            self._modules[str(i)] = m

    def __getitem__(self, idx):
        n = len(self._modules)
        if isinstance(idx, slice):
            return ModuleList(list(self._modules.values())[idx])
        if not -n <= idx < n:
            raise IndexError(f"index {idx} is out of range")
        return self._modules[str(idx % n)]

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())
