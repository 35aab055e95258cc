"""Network container + `net_guard` (tensorrt_llm/network.py:27-123).

The reference traces Module.forward into a TensorRT INetworkDefinition.  Here a Network simply records
which model was "traced" inside the guard and its named parameters; Builder.build_engine turns that into
the weight pack.  `plugin_config.set_identity_plugin` is accepted and ignored: the identity plugin only
exists to stop TensorRT from folding the KV outputs (README.md:100-106), and the in-place KV cache removes it.
"""
from __future__ import annotations

import contextlib
from typing import Optional

_ACTIVE = []


class PluginConfig:
    def __init__(self):
        self.identity_plugin = False

    def set_identity_plugin(self, dtype="float32"):
        self.identity_plugin = dtype
        return self

    def __getattr__(self, name):  # every other set_*_plugin of plugin/plugin.py:33-140 is a no-op here
        if name.startswith("set_"):
            return lambda *a, **k: self
        raise AttributeError(name)


class _TrtNetworkShim:
    name = ""


class Network:
    def __init__(self):
        self.trt_network = _TrtNetworkShim()  # build scripts set `network.trt_network.name`
        self.plugin_config = PluginConfig()
        self._named_parameters = None
        self.model = None
        self.inputs = None

    def set_named_parameters(self, named_parameters):
        self._named_parameters = list(named_parameters)

    def named_parameters(self):
        return iter(self._named_parameters or [])

    def _register_model(self, model):
        self.model = model


def default_net() -> Optional[Network]:
    return _ACTIVE[-1] if _ACTIVE else None


@contextlib.contextmanager
def net_guard(network: Network):
    _ACTIVE.append(network)
    try:
        yield network
    finally:
        _ACTIVE.pop()
