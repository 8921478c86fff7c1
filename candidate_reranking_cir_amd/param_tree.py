"""Build an nn.Module tree whose `state_dict()` has exactly a given key/shape layout.

The reference defines its parameters through a deep class hierarchy (vit.py, med.py,
nlvr_encoder.py); here the layout is data (`weights.*_param_spec`) and the tree is generated from
it, so `load_state_dict` / `state_dict` / `.parameters()` / `.to()` behave like the reference's
modules without restating their classes.
"""
from __future__ import annotations

import torch
from torch import nn

from .weights import synth_tensor


class ParamNode(nn.Module):
    """Container node (no forward): holds parameters / child nodes by state-dict name."""

    def extra_repr(self) -> str:
        return ", ".join(f"{k}{tuple(p.shape)}" for k, p in self._parameters.items())


def populate(root: nn.Module, spec, init_profile: str = "init", seed: int = 0) -> None:
    for key, (shape, kind) in spec.items():
        *path, leaf = key.split(".")
        node = root
        for name in path:
            if name not in node._modules:
                node.add_module(name, ParamNode())
            node = node._modules[name]
        value = synth_tensor(key, shape, kind, seed, init_profile)
        if kind == "position_ids":
            node.register_buffer(leaf, value)
        else:
            node.register_parameter(leaf, nn.Parameter(value, requires_grad=True))
